from fastdm_amd.sparse.config import RadialAttnConfig, SparseConfig  # noqa: F401
from fastdm_amd.sparse.xsparse import RadialAttn, SparseAttn  # noqa: F401
