from .config import get_config, make_config  # noqa: F401
