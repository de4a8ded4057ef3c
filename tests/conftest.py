import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_package():
    """The package directory is `slam-loop-closing_amd` (hyphen): load it by path as slam_loop_closing_amd."""
    name = "slam_loop_closing_amd"
    if name in sys.modules:
        return sys.modules[name]
    path = os.path.join(ROOT, "slam-loop-closing_amd")
    spec = importlib.util.spec_from_file_location(name, os.path.join(path, "__init__.py"),
                                                  submodule_search_locations=[path])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_oracle():
    name = "lcm_oracle_binding"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "oracle", "binding.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    mod.build()
    return mod


@pytest.fixture(scope="session")
def pkg():
    return load_package()


@pytest.fixture(scope="session")
def oracle():
    return load_oracle()


@pytest.fixture(scope="session")
def matcher(pkg):
    """One GPU matcher for the whole session (GPU tests run in one process)."""
    m = pkg.Matcher()
    yield m
    m.close()
