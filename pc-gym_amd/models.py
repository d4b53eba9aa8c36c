"""Host-side model metadata: the ``model.info()`` surface of the reference
(model_classes.py:8-20) for the models whose RHS exists as a HIP kernel.

Only names, orderings and default parameter values live here; the arithmetic is
in csrc/pcg_models.hpp.  Parameter order == declaration order of the reference
dataclass fields == order expected by the kernels (pcg_model_default_params()).
"""
from __future__ import annotations

from collections import OrderedDict

# ids must match enum pcg_model in include/pcgym_hip.h
CSTR, FOUR_TANK, ME, ME_REACTIVE, CRYST, AFFINE = range(6)


class ModelInfo:
    """Mirror of a reference model object as far as make_env needs it:
    ``info()`` -> {"parameters","states","inputs","disturbances"} and attribute
    access to parameters (pcgym.py:150-165, 226-238)."""

    def __init__(self, name, model_id, states, inputs, disturbances, params):
        self.name = name
        self.model_id = model_id
        self.states = list(states)
        self.inputs = list(inputs)
        self.disturbances = list(disturbances)
        self.parameters = OrderedDict(params)
        self.int_method = "hip"

    def info(self):
        return {
            "parameters": dict(self.parameters),
            "states": list(self.states),
            "inputs": list(self.inputs),
            "disturbances": list(self.disturbances),
        }

    def param_vector(self):
        return [float(v) for v in self.parameters.values()]

    def __getattr__(self, k):
        p = self.__dict__.get("parameters")
        if p is not None and k in p:
            return p[k]
        raise AttributeError(k)

    def copy(self):
        return ModelInfo(self.name, self.model_id, self.states, self.inputs, self.disturbances,
                         self.parameters)


def _registry():
    R = {}
    # model_classes.py:23-43
    R["cstr"] = ModelInfo(
        "cstr", CSTR, ["Ca", "T"], ["Tc"], ["Ti", "Caf"],
        [("q", 100.0), ("V", 100.0), ("rho", 1000.0), ("C", 0.239), ("deltaHr", -5e4),
         ("EA_over_R", 8750.0), ("k0", 7.2e10), ("UA", 5e4), ("Ti", 350.0), ("Caf", 1.0)])
    # model_classes.py:877-889, 924-926.  The reference lists ["None"] as a
    # disturbance (quirk Q13); it is unusable, we expose none.
    R["four_tank"] = ModelInfo(
        "four_tank", FOUR_TANK, ["h1", "h2", "h3", "h4"], ["v1", "v2"], [],
        [("g", 9.81), ("gamma_1", 0.2), ("gamma_2", 0.2), ("k1", 0.00085), ("k2", 0.00095),
         ("a1", 0.0035), ("a2", 0.0030), ("a3", 0.0020), ("a4", 0.0025),
         ("A1", 1.0), ("A2", 1.0), ("A3", 1.0), ("A4", 1.0)])
    # model_classes.py:361-367, 424-426
    R["multistage_extraction"] = ModelInfo(
        "multistage_extraction", ME,
        ["X1", "Y1", "X2", "Y2", "X3", "Y3", "X4", "Y4", "X5", "Y5"], ["L", "G"], ["X0", "Y6"],
        [("Vl", 5.0), ("Vg", 5.0), ("m", 1.0), ("Kla", 5.0), ("eq_exponent", 2.0),
         ("X0", 0.6), ("Y6", 0.05)])
    # model_classes.py:777-786, 857-859
    st = []
    for s in range(1, 6):
        st += [f"XA{s}", f"YA{s}", f"YB{s}", f"YC{s}"]
    R["multistage_extraction_reactive"] = ModelInfo(
        "multistage_extraction_reactive", ME_REACTIVE, st, ["L", "G"], [],
        [("Vl", 5.0), ("Vg", 5.0), ("m", 1.0), ("Kla", 0.01), ("k", 0.1), ("eq_exponent", 2.0),
         ("XA0", 2.0), ("YA6", 0.0), ("YB6", 2.0), ("YC6", 0.0)])
    # model_classes.py:1260-1270, 1340-1342.  The reference lists ka,kg,UA as
    # "disturbances" but the RHS ignores u[1:] (quirk Q13): none are wired.
    R["crystallization"] = ModelInfo(
        "crystallization", CRYST, ["Mu0", "Mu1", "Mu2", "Mu3", "Conc", "CV", "Ln"], ["Tc"], [],
        [("ka", 0.923714966), ("kb", -6754.878558), ("kc", 0.92229965554), ("kd", 1.341205945),
         ("kg", 48.07514464), ("k1", -4921.261419), ("k2", 1.871281405), ("a", 0.50523693),
         ("b", 7.271241375), ("alfa", 7.510905767), ("ro", 2.658)])
    return R


_REGISTRY = _registry()

# registry keys of the reference (pcgym.py:128-148) that are NOT built yet:
# asking for them is an explicit error, never a silent CPU fallback.
NOT_BUILT = [
    "complex_cstr", "first_order_system", "nonsmooth_control", "cstr_series_recycle",
    "distillation_column", "heat_exchanger", "biofilm_reactor", "polymerisation_reactor",
    "photobioreactor", "invariant_batch", "batch", "coupled_oscillator", "disease",
    "hydraulic_tank",
]


def get_model(name: str) -> ModelInfo:
    if name in _REGISTRY:
        return _REGISTRY[name].copy()
    if name in NOT_BUILT:
        raise ValueError(
            f"Model '{name}' exists in pc-gym but has no HIP kernel in this build "
            f"(built: {sorted(_REGISTRY)}).")
    # same message as the reference (pcgym.py:157)
    raise ValueError(f"Model '{name}' not found in model_mapping.")


def model_names():
    return sorted(_REGISTRY)
