"""The figures the head-group schedule sizes itself by (hybrid/async_attn_layer.py, comm/link.py): where each comes from
without a probe, how it is pinned, and that nothing collective happens unless asked for."""
import os

import pytest

import yunchang_amd.comm.link as L
import yunchang_amd.hybrid.async_attn_layer as AL


def test_fill_items_are_work_items_per_cu(monkeypatch):
    monkeypatch.setattr(AL, "_FILL_ITEMS", None)
    monkeypatch.setattr(L, "_cus", None)
    cus = L.device_cus()
    assert cus == 256                                   # no device in the CPU suite: the MI355X figure
    assert (AL.fill_items(), AL.fill_items(True), AL.fill_items(True, True)) == (2 * cus, cus, cus // 2)
    monkeypatch.setattr(L, "_cus", 304)                 # another part: the same per-CU rule
    assert (AL.fill_items(), AL.fill_items(True), AL.fill_items(True, True)) == (608, 304, 152)
    monkeypatch.setattr(AL, "_FILL_ITEMS", 1)           # the tests' pin wins everywhere
    assert AL.fill_items() == AL.fill_items(True) == AL.fill_items(True, True) == 1


def test_groups_follow_the_fill_rule(monkeypatch):
    monkeypatch.setattr(AL, "_FILL_ITEMS", None)
    monkeypatch.setattr(L, "_cus", 256)
    # 2 GPUs, ulysses 2: 16 heads -> 8 per rank; B2 S8192: 2 * 8 * 32 = 512 items per launch = ONE group at two items per CU,
    # two groups when the exchange is long against the attention, four when the forward kernel cuts along K as well
    assert AL._groups(16, 16, 2, 2, 8192)[0] == 1
    assert AL._groups(16, 16, 2, 2, 8192, link_bound=True)[0] == 2
    assert AL._groups(16, 16, 2, 2, 8192, link_bound=True, k_split=True)[0] == 4


def test_kernel_rate_constant_pin_and_use(monkeypatch):
    monkeypatch.setattr(L, "_kernel_measured", None)
    monkeypatch.delenv("USP_KERNEL_TFS", raising=False)
    assert L.kernel_flops_per_s() == L.DEFAULT_KERNEL_FLOPS_PER_S == 1.1e15 and not L.kernel_measured()
    monkeypatch.setenv("USP_KERNEL_TFS", "500")
    assert L.kernel_flops_per_s() == 5e14
    monkeypatch.delenv("USP_KERNEL_TFS")
    monkeypatch.setattr(L, "_kernel_measured", 2.2e15)   # what a probe would have stored
    assert L.kernel_flops_per_s() == 2.2e15 and L.kernel_measured()
    # _link_bound compares exchange time with HALF the attention time at that rate: a faster kernel makes the same exchange
    # link-bound sooner
    monkeypatch.setattr(AL, "_LINK_BYTES_PER_S", 64e9)
    monkeypatch.setattr(AL, "_KERNEL_FLOPS_PER_S", None)
    args = (16, 16, 2, 1, 16384, 128, 2, 1, True)        # Hq, Hkv, P, B, S, D, itemsize, ring, causal
    monkeypatch.setattr(L, "_kernel_measured", 0.2e15)
    slow = AL._link_bound(*args)
    monkeypatch.setattr(L, "_kernel_measured", 5e15)
    fast = AL._link_bound(*args)
    assert (slow, fast) == (False, True)


def test_no_probe_without_the_switch(monkeypatch):
    monkeypatch.delenv("USP_LINK_PROBE", raising=False)
    monkeypatch.setattr(L, "_measured", None)
    assert L.probe_link_rate(0, 8) is None and not L.measured()        # no process group exists: nothing collective was tried
    monkeypatch.setenv("USP_LINK_PROBE", "1")
    assert L.probe_link_rate(0, 1) is None                              # one rank: nothing to measure
