"""Same module paths as the reference package ``DiffusionFreeGuidence`` (its __init__.py star-imports the three modules)."""
from .DiffusionCondition import *   # noqa: F401,F403
from .ModelCondition import *       # noqa: F401,F403
from .TrainCondition import train, eval   # noqa: F401,E402
