from .config import CacheConfig, FBCacheConfig, TeaCacheConfig  # noqa: F401
from .xcaching import AutoCache, FBCache, TeaCache  # noqa: F401
