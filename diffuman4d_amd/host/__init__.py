"""Host-side mirror of the reference's sampler / pipeline / model interfaces (Python over the C ABI)."""
