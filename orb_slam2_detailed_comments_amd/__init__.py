"""MI355X-native ORB front-end (ORBextractor::operator() + Hamming matching) behind the C ABI of liborbx.so.

Python here is plumbing only: ctypes marshalling, torch.distributed sharding of frame batches, synthetic
inputs.  All compute is in csrc/ (hand-written HIP for gfx950).  There is no CPU fallback.
"""
from ._capi import KP_DTYPE, OrbxError, LIB_PATH  # noqa: F401
from .extractor import ORBextractor  # noqa: F401
from .matcher import ORBmatcher  # noqa: F401
from .frame import Frame  # noqa: F401
from .vocabulary import ORBVocabulary  # noqa: F401
