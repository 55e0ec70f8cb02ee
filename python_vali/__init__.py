"""Drop-in alias: `import python_vali as vali` resolves to the MI355X-native package."""
from vali_amd import *  # noqa: F401,F403
from vali_amd import __version__  # noqa: F401
import vali_amd as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("_")})
