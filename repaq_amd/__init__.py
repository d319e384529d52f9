"""repaq_amd — MI355X (gfx950) engine for the OpenGene/repaq RfqCodec path: FASTQ <-> .rfq, bit-identical to repaq v0.5.1.

All compute is hand-written HIP behind the C-ABI of include/rfq_hip.h (repaq_amd/lib/librfq_hip.so).  This package is
the thin host-side mirror of the reference's RfqCodec interface; it has no CPU implementation."""
from .codec import RfqCodec, RfqError, SE, PE_TWO_FILES, PE_INTERLEAVED, nolb_threshold  # noqa: F401

__all__ = ["RfqCodec", "RfqError", "SE", "PE_TWO_FILES", "PE_INTERLEAVED", "nolb_threshold"]
