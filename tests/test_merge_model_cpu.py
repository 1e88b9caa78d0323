"""The host model of merge_runs_lds (tools/model_prefix_merge.py): the index arithmetic of the in-LDS merge of ascending runs —
pair tables, chunk assignment, the gapped layout with one sentinel cell per group, carried groups, the checked first round —
executed thread by thread.  Keeps the scheme the kernels implement pinned on the CPU."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_merge_model_sorts_every_run_structure():
    spec = importlib.util.spec_from_file_location("model_prefix_merge", os.path.join(ROOT, "tools", "model_prefix_merge.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.main(cases=600)
