from .config import CacheConfig, TeaCacheConfig  # noqa: F401
from .xcaching import AutoCache, TeaCache  # noqa: F401
