"""tasks/tomospc_cabinet.py of the reference: see tomodj_cabinet.py (one shared mirror)."""
from .tomodj_cabinet import TomoSPCCabinetTask  # noqa: F401
