from .config import CacheConfig, DiCacheConfig, FBCacheConfig, TeaCacheConfig  # noqa: F401
from .xcaching import AutoCache, DiCache, FBCache, TeaCache  # noqa: F401
