"""MI355X-native Whisper-Medusa inference engine — drop-in for the reference package name.

``from whisper_medusa import WhisperMedusaModel`` works exactly as with
aiola-lab/whisper-medusa (README.md:101-142); the hot path runs in ``libwm.so``
(hand-written HIP for gfx950, ``csrc/``) behind the C-ABI of ``include/wm.h``.
"""
from .config import MedusaConfig, GenParams, ACCEPT_TYPICAL, ACCEPT_GREEDY  # noqa: F401


def __getattr__(name):
    # lazy: importing the config/synth helpers must not require the HIP library
    if name in ("WhisperMedusaModel", "get_model"):
        from . import api
        return getattr(api, name)
    raise AttributeError(name)
