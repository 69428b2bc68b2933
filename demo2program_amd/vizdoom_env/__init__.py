"""ViZDoom side of the training path: DSL vocabulary / parser / canonical form and the dataset
reader.  The game engine (vizdoom_env/vizdoom_env.py in the reference) is not part of this build."""
from .dsl import VizDoomDSLVocab, parse  # noqa: F401
