import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    """-m gpu runs: every RolloutEngine.step / reset first fills the LDS of all CUs with 0xFF bytes (fmarl_poison_lds), so a
    kernel that reads an LDS table before writing it fails deterministically instead of seeing the values its previous launch
    left there (an uninitialised word of the formation kernel at N = 1 once survived a whole round that way).
    FMARL_TEST_POISON=0 switches it off."""
    if os.environ.get('FMARL_TEST_POISON', '1') == '0' or not any(i.get_closest_marker('gpu') for i in items):
        return
    try:
        import torch
        if not torch.cuda.is_available():
            return
        from fair_marl_amd.engine import RolloutEngine
    except Exception:
        return
    if getattr(RolloutEngine, '_poisoned', False):
        return
    def poisoned(fn):
        def call(self, *a, **kw):
            if not torch.cuda.is_current_stream_capturing():
                self.poison_lds()
            return fn(self, *a, **kw)
        call.__name__, call.__doc__ = fn.__name__, fn.__doc__
        return call
    for name in ('step', 'step_span', 'reset', 'rebuild_graph', 'update_graph', 'process_adj', 'process_infos', 'lexifair', 'cost_matrix'):
        if hasattr(RolloutEngine, name):
            setattr(RolloutEngine, name, poisoned(getattr(RolloutEngine, name)))
    RolloutEngine._poisoned = True
