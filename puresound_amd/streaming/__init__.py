"""Mirror of `puresound.streaming` (chunked / frame-by-frame forward of the SkiM masker)."""
from .skim_inference import StreamingSkiM  # noqa: F401
