"""Kernel-instantiation coverage of the running pytest session (filled by conftest.py, read by test_zz_kernel_coverage.py)."""
COVERAGE = {}  # mangled kernel name -> {"oracle": [test ids], "golden": [...], "other": [...]}
COVERAGE_STATE = {"gpu_deselected": 0, "gpu_ran": 0, "gpu_failed": 0}
