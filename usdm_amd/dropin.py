"""Makes the reference's import paths resolve to the MI355X implementations:

    import usdm_amd.dropin; usdm_amd.dropin.install()
    from voicebox.util.model_util import initialize_decoder, reconstruct_speech     # src/inference.py:12
    from voicebox.model import Voicebox; from voicebox.vocoder.models import BigVGAN  # model_util.py:12-15
    from seamless_communication.models.unit_extractor import UnitExtractor            # src/inference.py:9
"""
import importlib
import sys
import types


def install():
    from . import unit_extractor, voicebox
    sys.modules["voicebox"] = voicebox
    for sub in ("model", "model.voicebox", "model.networks", "util", "util.model_util", "vocoder", "vocoder.models", "vocoder.env"):
        sys.modules["voicebox." + sub] = importlib.import_module("usdm_amd.voicebox." + sub)
    for name in ("seamless_communication", "seamless_communication.models"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["seamless_communication.models.unit_extractor"] = unit_extractor
    sys.modules["seamless_communication"].models = sys.modules["seamless_communication.models"]
    sys.modules["seamless_communication.models"].unit_extractor = unit_extractor
