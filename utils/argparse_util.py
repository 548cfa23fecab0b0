from ndivplanning_amd.utils.argparse_util import *  # noqa: F401,F403
from ndivplanning_amd.utils.argparse_util import override_dotmap  # noqa: F401
