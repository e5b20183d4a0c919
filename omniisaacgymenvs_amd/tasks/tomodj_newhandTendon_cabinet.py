"""tasks/tomodj_newhandTendon_cabinet.py of the reference: see newhand_cabinet.py (one shared mirror)."""
from .newhand_cabinet import TomoDJnewhandTendonCabinetTask  # noqa: F401
