"""aerobulk_amd — MI355X-native bulk air-sea flux engine (drop-in for AeroBulk's aerobulk_compute path).

Host mirror of the reference interface lives in `aerobulk_amd.api`; the product is the HIP
library `libaerobulk_amd.so` (C ABI: include/aerobulk_amd.h)."""
from .api import (ALGOS, AerobulkError, Session, aerobulk_model, calibrate, device_count, phymbl, synth_fields_device, turb_ice,  # noqa: F401
                  turb_neutral_10m)

__all__ = ["ALGOS", "AerobulkError", "Session", "aerobulk_model", "calibrate", "device_count", "phymbl", "synth_fields_device", "turb_ice", "turb_neutral_10m"]
