/* usp_item_deal.h -- which (head, tile) a work-item id of an XCD's run stands for.  Plain C: included by the HIP
 * kernels (usp_common.hpp: ItemWalk::dealt) and compiled as-is by tests/test_host_api.py, which checks on the host that
 * the map is a bijection and that the eight runs weigh the same for every head / tile count it is meant for.
 *
 * A launch's item list is cut into 8 contiguous runs of items_l ids, one per XCD (all sharers of one K/V behind one
 * L2).  id = head * n_inner + tile, tile 0 the heaviest (causal).  Fewer than 8 heads (the Ulysses-8 / small
 * head-group case): one head has more tiles than an XCD has items, and cut contiguously its tiles give one XCD the
 * heavy end of the causal triangle and the next the light end (B1 H4/1 S32768: 1008 TFLOP/s forward against 1190
 * dealt; backward 655 against 824).  So they are dealt.  Regular case, n_inner = m * items_l: the m XCDs of a head
 * take its tiles round by round in alternating direction (XCD j: tile i*m + j, or i*m + m-1-j in odd rounds).
 * Irregular head counts (3, 5, 6, 7): every head is dealt to all 8 XCDs the same way and the run is ordered
 * round-major, i.e. still heaviest first across the heads (B1 H6 S16384 forward 769 -> 1020).  Every share is sorted
 * heaviest first and all shares weigh the same.  Launches with 8 or more heads keep the contiguous runs (dealing the
 * remainder heads of 12 or 20 measured -1...-5 %: what they gain in balance they lose in K/V sharers behind one L2).
 *
 * w: id in run order (run x = w / items_l); returns the id to decode. */
#ifndef USP_ITEM_DEAL_H
#define USP_ITEM_DEAL_H

#ifndef USP_DEAL_FN
#define USP_DEAL_FN static inline
#endif

USP_DEAL_FN int usp_deal_item(int w, int n_inner, int items_l) {
  if (items_l >= n_inner) return w;               /* whole heads per XCD first (always so without the XCD split) */
  const int x = w / items_l, loc = w - x * items_l;
  if (n_inner % items_l == 0) {
    const int m = n_inner / items_l, j = x % m;
    return (x / m) * n_inner + loc * m + ((loc & 1) ? (m - 1 - j) : j);
  }
  if ((n_inner & 7) != 0) return w;
  {
    const int heads = items_l / (n_inner >> 3);   /* items_l = heads * n_inner / 8 */
    const int i = loc / heads, head = loc - i * heads;
    return head * n_inner + i * 8 + ((i & 1) ? (7 - x) : x);
  }
}

/* Round 6: the query heads of ONE KV group side by side.  An XCD's run that holds several whole heads is walked head-major by the
 * ids above: its 32 workgroups stream the K / V of one query head's tiles, then the same K / V again for the next head of the
 * group.  Here the run's ids are re-read tile-major inside blocks of m = min(heads per run, G) heads (m | G, so a block never
 * leaves its KV group): consecutive ids -- what the XCD's workgroups hold at any one time -- are the same tile of m heads, which
 * read the same K / V rows through the same L2.  Still a bijection, still heaviest first, the alternating passes still pair equal
 * weights (tests/test_host_api.py).  w: id in run order, after usp_deal_item (which only deals when a head spans several runs:
 * then this map is the identity). */
USP_DEAL_FN int usp_group_item(int w, int n_inner, int items_l, int G) {
  if (G <= 1 || items_l < 2 * n_inner || items_l % n_inner != 0) return w;
  {
    const int hr = items_l / n_inner;             /* whole heads per run */
    const int m = hr < G ? hr : G;
    if (G % m != 0 || hr % m != 0) return w;
    {
      const int x = w / items_l, loc = w - x * items_l;
      const int blk = loc / (m * n_inner), j = loc - blk * (m * n_inner);
      return (x * hr + blk * m + j % m) * n_inner + j / m;
    }
  }
}


#endif
