"""Import shim: ``import cgc_net_amd`` loads the package that lives in ``cgc-net_amd/``.

The package directory carries the project's name (with a hyphen), which Python
cannot import directly; this module replaces itself in ``sys.modules`` with the
real package so that ``cgc_net_amd.network`` etc. resolve into that directory.
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'cgc-net_amd')
_spec = importlib.util.spec_from_file_location(
    'cgc_net_amd', os.path.join(_dir, '__init__.py'), submodule_search_locations=[_dir])
_pkg = importlib.util.module_from_spec(_spec)
sys.modules['cgc_net_amd'] = _pkg
_spec.loader.exec_module(_pkg)
