"""Importable alias of the `filmyou-core_amd/` package (a hyphen is not a valid Python identifier)."""
import importlib
import sys

sys.modules[__name__] = importlib.import_module("filmyou-core_amd")
