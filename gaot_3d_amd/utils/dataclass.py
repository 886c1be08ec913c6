"""Dataclass helpers with the semantics of the reference's src/utils/dataclass.py:5-24 (only declared
dataclass FIELDS are forwarded -- attributes attached later with setattr are ignored), minus omegaconf."""
from dataclasses import fields, is_dataclass
from typing import Any


def shallow_asdict(obj: Any) -> dict:
    if is_dataclass(obj):
        return {f.name: getattr(obj, f.name) for f in fields(obj)}
    if isinstance(obj, dict):
        return dict(obj)
    raise TypeError(f"Unsupported type for shallow_asdict: {type(obj)}")


def safe_replace(obj: Any, **kwargs) -> Any:
    if is_dataclass(obj):
        names = {f.name for f in fields(obj)}
        for k, v in kwargs.items():
            if k in names:
                setattr(obj, k, v)
        return obj
    if isinstance(obj, dict):
        obj.update(kwargs)
        return obj
    raise TypeError(f"Unsupported type for safe_replace: {type(obj)}")
