"""malio_amd - MI355X-native measurement-update engine for MA-LIO (hot path only).

The directory is named `ma-lio_amd` (not importable as-is); load it with
`__graft_entry__.load_package()` which registers it as the module `malio_amd`.
"""
from . import scenes  # noqa: F401
