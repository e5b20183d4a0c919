"""tasks/tomospc_newhand_cabinet.py of the reference: see newhand_cabinet.py (one shared mirror)."""
from .newhand_cabinet import TomoSPCnewhandCabinetTask  # noqa: F401
