"""MI355X-native (gfx950) implementation of the MDGAT-matcher inference hot path.

``MDGAT`` is a drop-in for ``models.mdgat.MDGAT`` of the reference (FPFH descriptor, inference);
``match`` is the functional convenience API.  All arithmetic runs in ``libmdgat_hip.so``
(hand-written HIP, C ABI in ``include/mdgat_hip.h``)."""
from .mdgat import MDGAT, match  # noqa: F401

__all__ = ['MDGAT', 'match']
