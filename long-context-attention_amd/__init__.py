"""MI355X-native Unified Sequence Parallel attention.

Drop-in for the hot path of feifeibear/long-context-attention ("yunchang" 0.6.4):
    set_seq_parallel_pg -> EXTRACT_FUNC_DICT -> LongContextAttention(ring_impl_type="zigzag")
with the same names, arguments and error behaviour (yunchang/__init__.py:1-12).  The directory is
named long-context-attention_amd; import it as `yunchang_amd` (see yunchang_amd/__init__.py).
"""
from .hybrid import *  # noqa: F401,F403
from .ring import *  # noqa: F401,F403
from .ulysses import *  # noqa: F401,F403
from .globals import set_seq_parallel_pg, PROCESS_GROUP
from .comm.extract_local import (
    stripe_extract_local,
    basic_extract_local,
    zigzag_extract_local,
    EXTRACT_FUNC_DICT,
)
from .kernels import AttnType, select_flash_attn_impl

__version__ = "0.6.0"
