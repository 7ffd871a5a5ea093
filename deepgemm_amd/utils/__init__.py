from . import math, layout
from .layout import *     # noqa: F401,F403
from .math import *       # noqa: F401,F403
from .dist import init_dist, uneven_all_gather, dist_print    # noqa: F401
