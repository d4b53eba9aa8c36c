#!/usr/bin/env python3
"""The Rosenbrock instance of tools/integrator_sweep.py (the name pcg_integrators.hpp's comment on ros_try_rolled cites):
every registry model x {rodas3, rodas4, rodas5} x counter mode through the general step kernel against the oracle."""
import os
import runpy
import sys

sys.argv = [sys.argv[0]] + (sys.argv[1:] or ["rodas3", "rodas4", "rodas5"])
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "integrator_sweep.py"), run_name="__main__")
