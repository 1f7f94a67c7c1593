"""fmhr_amd.examples — caller-side examples of the drop-in surface (not used by the product path)."""
