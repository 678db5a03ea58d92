"""Build-container-only shim so the upstream reference (read-only, /root/reference) imports under
numpy 2 / no Postgres.  Test infrastructure: never imported by the product path.  See SURVEY.md App. A."""
import sys
import types

import numpy

if not hasattr(numpy, "infty"):
    numpy.infty = numpy.inf  # removed in numpy 2 (reference: rl4mm/orderbook/Exchange.py:154)


def _fake_postgres_engine():
    import sqlalchemy

    mod = types.ModuleType("rl4mm.database.PostgresEngine")
    mod.MAX_POOL_SIZE = 30

    class PostgresEngine:  # replaces rl4mm/database/PostgresEngine.py:19-29 with sqlite in RAM
        def __init__(self, *a, **k):
            self.engine = sqlalchemy.create_engine("sqlite:///:memory:")

    mod.PostgresEngine = PostgresEngine
    return mod


class _Finder:
    """Serve the fake PostgresEngine module the first time anything imports it."""

    def find_spec(self, name, path=None, target=None):
        if name != "rl4mm.database.PostgresEngine":
            return None
        import importlib.machinery

        class _Loader:
            def create_module(self, spec):
                return _fake_postgres_engine()

            def exec_module(self, module):
                pass

        return importlib.machinery.ModuleSpec(name, _Loader())


sys.meta_path.insert(0, _Finder())
