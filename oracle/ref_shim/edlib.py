"""edlib stub so that the reference's barcode_trimmer imports (out of scope, never called)."""
def align(*a, **k):
    raise RuntimeError("edlib is not available in this build (barcode trimming is out of scope)")
