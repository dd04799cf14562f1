"""Importable alias of the package directory ``quantizedneuralnetworks-keras-tensorflow_amd``
(whose name is not a Python identifier)."""
import importlib
import sys

_pkg = importlib.import_module("quantizedneuralnetworks-keras-tensorflow_amd")
sys.modules[__name__] = _pkg
