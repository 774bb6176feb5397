from .all_to_all import SeqAllToAll4D, SeqAllToAll5D, all_to_all_4D, all_to_all_5D
from .extract_local import (EXTRACT_FUNC_DICT, basic_extract_local, stripe_extract_local,
                            zigzag_extract_local)

__all__ = ["SeqAllToAll4D", "SeqAllToAll5D", "all_to_all_4D", "all_to_all_5D", "EXTRACT_FUNC_DICT",
           "basic_extract_local", "stripe_extract_local", "zigzag_extract_local"]
