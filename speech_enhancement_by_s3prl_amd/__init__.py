"""Importable alias of the ``speech-enhancement-by-s3prl_amd/`` directory (a hyphenated name cannot be
imported): submodules resolve there through ``__path__``.

    from speech_enhancement_by_s3prl_amd import preprocessor, heads, transformer, objective, decode
"""
import os as _os

__path__.append(_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                              'speech-enhancement-by-s3prl_amd'))
__version__ = '0.1'
