from . import io  # noqa: F401
