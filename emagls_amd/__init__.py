"""emagls_amd -- MI355X-native eMagLS filter design and binaural rendering (HIP, gfx950).

The Python layer mirrors the reference's MATLAB entry points over the C ABI in include/emagls.h.
Importing the package does not need a GPU; calling any function needs emagls_amd/lib/libemagls.so
(python -m emagls_amd.build) and an MI355X -- there is no CPU fallback.
"""
from .api import (applyRadialFilter, binauralDecode, designHrirSets, encodeSH, fromAtfHrirSets, getEMagLs2Filters, getEMagLsFilters, getEMagLsFiltersEMAinCH, getEMagLsFiltersEMAinSH,
                  getEMagLsFiltersFromAtf, getLsFilters, getMagLsArrayDiffuseFilter, getMagLsFilters, getMagLsFilters2D,
                  getMagLsSphericalHeadFilter, getCH, getRadialFilter, getSH, getSMAIRMatrix, sphModalCoeffs)
from .plan import Batch, Plan

__all__ = ["getLsFilters", "getMagLsFilters", "getEMagLsFilters", "getEMagLs2Filters", "getEMagLsFiltersEMAinCH", "getEMagLsFiltersEMAinSH", "getEMagLsFiltersFromAtf",
           "binauralDecode", "getSH", "getCH", "getSMAIRMatrix", "sphModalCoeffs", "getMagLsFilters2D", "getRadialFilter", "applyRadialFilter", "encodeSH",
           "getMagLsSphericalHeadFilter", "getMagLsArrayDiffuseFilter", "designHrirSets", "fromAtfHrirSets", "Plan", "Batch"]
