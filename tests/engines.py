"""Engine selection for the parity tests: the same test bodies run against
  * "emu": the lane-loop emulation build of the kernel source (CPU, every run)
  * "hip": the product, libcorintho_hip.so on a real MI355X (-m gpu)
"""
import pytest

from corintho_ai_amd import trainer as T

ENGINES = [pytest.param("emu", id="emu"), pytest.param("hip", marks=pytest.mark.gpu, id="hip")]


def cdll(name):
    if name == "emu":
        from tests.emu import emulib

        return emulib.load()
    from corintho_ai_amd import _lib

    return _lib.load()


def make_trainer(name, *args, **kw):
    return T.Trainer(*args, _cdll=cdll(name), **kw)
