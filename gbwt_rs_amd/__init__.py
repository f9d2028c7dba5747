"""gbwt_rs_amd -- MI355X-native (gfx950, hand-written HIP) drop-in for the GBWT LF-step hot path of
jltsiren/gbwt-rs: batched path extraction (SequenceIter) and find/extend/bidirectional search.

`api`   : host-side mirror of the reference's GBWT / GBZ interface over the C ABI (include/gbwt_hip.h)
`synth` : synthetic GBWT/GBZ generator and simple-sds writer (host only)
`csrc`  : HIP kernels + C ABI (libgbwt_hip.so)
"""
from ._lib import OPEN_ALL, OPEN_EXTRACT, OPEN_GFA, OPEN_SEARCH
from .api import (BD_DTYPE, FORWARD, GBWT, GBZ, PATHS_DEFAULT, PATHS_PAN_SN, PATHS_REF_ONLY, POS_DTYPE, REVERSE, STATE_DTYPE, GbwtHipError, decode_node, device_count, device_memory,
                  encode_node, encode_path, flip_node, parse_file)

__all__ = ["OPEN_ALL", "OPEN_EXTRACT", "OPEN_GFA", "OPEN_SEARCH", "GBWT", "GBZ", "GbwtHipError", "FORWARD", "REVERSE", "PATHS_DEFAULT", "PATHS_PAN_SN", "PATHS_REF_ONLY", "POS_DTYPE", "STATE_DTYPE", "BD_DTYPE", "encode_node",
           "decode_node", "flip_node", "encode_path", "device_count", "device_memory", "parse_file"]
