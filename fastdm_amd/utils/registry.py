"""JSON -> dataclass loading through a name registry.

Both configuration families of the reference (fastdm/caching/config.py, fastdm/sparse/config.py) work the same way: a
base dataclass with the common fields, sub-dataclasses registered under an algorithm name, `from_dict` picking the
sub-class by a selector key and silently dropping keys the target does not declare (the shipped JSON files rely on
that), `from_json` reading a file first.  This module implements that behaviour once.
"""
import dataclasses
import json
import os
from typing import Any, Dict


class RegisteredConfig:
    """Mixin for a base @dataclass.  Set `_selector` (the JSON key naming the algorithm) on the base class."""
    _selector = "algorithm"
    _table = None          # one dict per family, created by the first register() call on that family's base class

    @classmethod
    def _family(cls):
        for klass in cls.__mro__:
            if "_selector" in klass.__dict__:
                return klass
        raise TypeError(f"{cls.__name__} does not belong to a configuration family")

    @classmethod
    def register(cls, name: str):
        fam = cls._family()
        if fam.__dict__.get("_table") is None:
            fam._table = {}

        def wrap(sub):
            fam._table[name.lower()] = sub
            return sub
        return wrap

    @classmethod
    def from_dict(cls, data: Dict[str, Any]):
        fam = cls._family()
        algo = data.get(fam._selector)
        if algo is None:
            raise ValueError(f"{fam._selector} must be specified in the config data.")
        target = (fam.__dict__.get("_table") or {}).get(str(algo).lower(), cls)
        known = {f.name for f in dataclasses.fields(target)}
        return target(**{key: value for key, value in data.items() if key in known})

    @classmethod
    def from_json(cls, file_path: str):
        if not os.path.exists(file_path):
            raise FileNotFoundError(f"file is not exist: {file_path}")
        with open(file_path, "r", encoding="utf-8") as handle:
            return cls.from_dict(json.load(handle))
