from .dataclass import shallow_asdict, safe_replace  # noqa: F401
