"""Import location kept for code written against ``rlpyt.envs.base``; see ``rlpyt_amd.envs``."""
from . import Env, EnvInfo, EnvSpaces, EnvStep  # noqa: F401
