"""fastdm_amd sub-package (mirrors the reference package layout so import paths carry over)."""
