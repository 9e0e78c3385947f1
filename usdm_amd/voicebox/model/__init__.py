from .voicebox import Voicebox  # noqa: F401
