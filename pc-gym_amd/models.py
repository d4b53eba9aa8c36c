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
COMPLEX_CSTR, DISEASE, BATCH, PHOTO, CSTR_SERIES, DISTILLATION, POLYMER = range(6, 13)
BIOFILM, HEAT_EX, INV_BATCH, OSCILLATORS = range(13, 17)
USER = 17  # custom_model with a C-expression right-hand side, compiled at plan creation (config.py)


class ModelInfo:
    """Mirror of a reference model object as far as make_env needs it:
    ``info()`` -> {"parameters","states","inputs","disturbances"} and attribute
    access to parameters (pcgym.py:150-165, 226-238)."""

    def __init__(self, name, model_id, states, inputs, disturbances, params, affine_builder=None):
        self.affine_builder = affine_builder  # registry models whose RHS is affine: params -> (A, B, c)
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
        return type(self)(self.name, self.model_id, self.states, self.inputs, self.disturbances,
                         self.parameters, self.affine_builder)


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
    # ---- "next" row f-2: further registry models (general kernels) --------------------------------
    # model_classes.py:65-96
    R["complex_cstr"] = ModelInfo(
        "complex_cstr", COMPLEX_CSTR, ["Ca", "Cb", "Cc", "T"], ["Tc"], ["Ti", "Caf"],
        [("q", 100.0), ("V", 100.0), ("rho", 1000.0), ("C", 0.239), ("deltaHr1", -5e4), ("EA1_over_R", 8750.0),
         ("k01", 7.2e10), ("deltaHr2", -3e4), ("EA2_over_R", 9000.0), ("k02", 1.0e10), ("UA", 5e4),
         ("Ti", 350.0), ("Caf", 1.0)])
    # model_classes.py:156-171 (registry key "disease", pcgym.py:146)
    R["disease"] = ModelInfo("disease", DISEASE, ["S", "I", "R"], ["u"], [], [("beta", 0.3), ("gamma", 0.1)])
    # model_classes.py:222-246
    R["batch"] = ModelInfo(
        "batch", BATCH, ["Ca", "Cb", "Cc", "T"], ["Tc"], [],
        [("k01", 1.0), ("k02", 0.5), ("EA1", 5000.0), ("EA2", 6000.0), ("R", 8.314), ("dH1", -1000.0),
         ("dH2", -1500.0), ("rho", 1000.0), ("Cp", 4.0), ("UA", 100.0), ("V", 1.0)])
    # model_classes.py:443-453, 495-503 (registry key "photobioreactor", pcgym.py:141)
    R["photobioreactor"] = ModelInfo(
        "photobioreactor", PHOTO, ["c_x", "c_N", "c_q"], ["I", "F_N"], [],
        [("u_m", 0.0572), ("u_d", 0.0), ("Y_NX", 504.5), ("k_m", 0.00016), ("k_d", 0.281), ("k_sq", 23.51),
         ("K_Nq", 16.89), ("k_iq", 800.0), ("k_s", 178.9), ("k_i", 447.1), ("k_N", 393.1)])
    # model_classes.py:619-630, 672-677
    R["cstr_series_recycle"] = ModelInfo(
        "cstr_series_recycle", CSTR_SERIES, ["C1", "T1", "C2", "T2"], ["F", "L", "Tc1", "Tc2"], [],
        [("C_O", 97.35), ("T_O", 298.0), ("V1", 1e-3), ("V2", 2e-3), ("U1A1", 0.461), ("U2A2", 0.732),
         ("rho", 1.05e3), ("cp", 3.766), ("k", 3.118e5), ("E", 46.14), ("deltaH", 58.41), ("R", 8.3145e-3)])
    # model_classes.py:689-695, 753-758
    R["distillation_column"] = ModelInfo(
        "distillation_column", DISTILLATION, ["X0", "X1", "X2", "X3", "Xf", "X4", "X5", "X6", "Xb"], ["R", "F"], [],
        [("D", 100.0), ("q", 1.0), ("alpha", 5.0), ("X_feed", 0.2), ("M0", 2000.0), ("Mb", 2000.0), ("M", 2000.0)])
    # model_classes.py:1172-1182, 1222-1227
    R["polymerisation_reactor"] = ModelInfo(
        "polymerisation_reactor", POLYMER, ["T", "M", "I"], ["F", "Tf", "Mf", "If"], [],
        [("Ap", 6e10), ("Ad", 4e10), ("At", 9e10), ("Ep_over_R", 7750.0), ("Ed_over_R", 8500.0),
         ("Et_over_R", 8250.0), ("f", 0.5), ("V", 1.0), ("deltaHp", -3e4), ("rho", 1200.0), ("cp", 2.0)])
    # model_classes.py:1062-1073, 1148-1150
    st = []
    for s in ("1", "2", "3", "A"):
        st += [f"S1_{s}", f"S2_{s}", f"S3_{s}", f"O_{s}"]
    R["biofilm_reactor"] = ModelInfo(
        "biofilm_reactor", BIOFILM, st, ["F", "Fr", "S1_F", "S2_F", "S3_F"], [],
        [("V", 10.0), ("Va", 15.0), ("Kla", 1.5), ("m", 0.5), ("eq_exponent", 1.0), ("O_air", 300.0), ("vm_1", 0.8),
         ("vm_2", 1.0), ("K1", 0.5), ("K2", 0.1), ("KO_1", 1.5), ("KO_2", 0.5)])
    # model_classes.py:949-960, 1040-1041 (its info() has no "disturbances" key)
    st = []
    for s in range(1, 9):
        st += [f"Tt{s}", f"Tm{s}", f"Ts{s}"]
    R["heat_exchanger"] = ModelInfo(
        "heat_exchanger", HEAT_EX, st, ["Ft", "Fs", "Tt0", "Ts9"], [],
        [("Utm", 1.0), ("Usm", 1.0), ("L", 1.0), ("Dt", 1.0), ("Dm", 2.0), ("Ds", 3.0), ("cpt", 1.0), ("cpm", 1.0),
         ("cps", 1.0), ("rhot", 1.0), ("rhom", 1.0), ("rhos", 1.0)])
    # model_classes.py:268-293 (no inputs)
    R["invariant_batch"] = ModelInfo(
        "invariant_batch", INV_BATCH, ["xA", "xB", "xC", "xD"], [], [],
        [("k1f", 55.0), ("k1r", 1.0), ("k2f", 2.0), ("k2r", 1.0)])
    # model_classes.py:186-216 (no inputs; N is structural: only the default ring of 10 masses is compiled)
    R["coupled_oscillator"] = ModelInfo(
        "coupled_oscillator", OSCILLATORS, [f"x{i + 1}" for i in range(10)] + [f"p{i + 1}" for i in range(10)], [], [],
        [("N", 10), ("k", 1.0), ("m", 1.0)])
    # registry models whose RHS is affine run on the affine kernel (matrices built here from the parameters)
    # hydraulic_tank model_classes.py:128-153: dq1 = -D (q1-q2) + u, dq2 = D (q1-q2) - u
    R["hydraulic_tank"] = ModelInfo(
        "hydraulic_tank", AFFINE, ["q1", "q2"], ["u"], [], [("D", 1.0)],
        affine_builder=lambda p: ([[-p["D"], p["D"]], [p["D"], -p["D"]]], [[1.0], [-1.0]], [0.0, 0.0]))
    # first_order_system model_classes.py:296-343: dx = (K u - x) / tau   ("None" disturbance entry: quirk Q13)
    R["first_order_system"] = ModelInfo(
        "first_order_system", AFFINE, ["x"], ["u"], [], [("K", 1.0), ("tau", 0.5)],
        affine_builder=lambda p: ([[-1.0 / p["tau"]]], [[p["K"] / p["tau"]]], [0.0]))
    # nonsmooth_control model_classes.py:509-558
    R["nonsmooth_control"] = ModelInfo(
        "nonsmooth_control", AFFINE, ["X1", "X2"], ["U"], [],
        [("a_11", 0.0), ("a_12", 1.0), ("a_21", -2.0), ("a_22", -3.0), ("b_1", 0.0), ("b_2", 1.0)],
        affine_builder=lambda p: ([[p["a_11"], p["a_12"]], [p["a_21"], p["a_22"]]], [[p["b_1"]], [p["b_2"]]], [0.0, 0.0]))
    return R


_REGISTRY = _registry()

# every registry key of the reference (pcgym.py:128-148) is built; kept for the explicit-error path
NOT_BUILT = []


# class names of the reference's model objects (pcgym.py:128-148): callers inspect env.model.__class__.__name__
# (tests/environment/test_make_env_basic.py:26)
_CLASS_NAME = {"photobioreactor": "photo_production", "disease": "disease_model",
               "coupled_oscillator": "coupled_oscillators"}
_CLASS_CACHE = {}


def _named(mi: ModelInfo) -> ModelInfo:
    cname = _CLASS_NAME.get(mi.name, mi.name)
    cls = _CLASS_CACHE.setdefault(cname, type(cname, (ModelInfo,), {}))
    mi.__class__ = cls
    return mi


def get_model(name: str) -> ModelInfo:
    if name in _REGISTRY:
        return _named(_REGISTRY[name].copy())
    if name in NOT_BUILT:  # pragma: no cover - empty since every registry model has a kernel
        raise ValueError(
            f"Model '{name}' exists in pc-gym but has no HIP kernel in this build "
            f"(built: {sorted(_REGISTRY)}).")
    # same message as the reference (pcgym.py:157)
    raise ValueError(f"Model '{name}' not found in model_mapping.")


def model_names():
    return sorted(_REGISTRY)
