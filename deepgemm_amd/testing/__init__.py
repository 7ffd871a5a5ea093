from . import bench, numeric, utils, generators
from .bench import *      # noqa: F401,F403
from .numeric import *    # noqa: F401,F403
from .utils import *      # noqa: F401,F403
