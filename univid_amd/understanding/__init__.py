"""Understanding path of UniVid (BASELINE.json config 5): the SigLIP2 frame ranker of the Pyramid-Reflection loop."""
from .eval_understanding import Siglip2Scorer, mmr_select  # noqa: F401
from .siglip2 import Siglip2Model  # noqa: F401
