"""``SpciesOptions`` - host-side mirror of the reference's ``classes/Spcies_options.m``.

Same option names, same precedence (explicit name-value > ``options`` struct > the plugin's
``def_options_*`` > class constants, ``Spcies_options.m:139-272``), same accepted
formulation/method/submethod tables (``:63-106``) and the same routing of unknown fields into
``.solver`` (``:556-603``).  The one addition is the ``'HIP'`` platform in ``valid_platform``
(``:65`` lists only ``'C'`` and ``'Matlab'``): that is where a new back-end is admitted.
"""
from __future__ import annotations

import copy
import warnings
from types import SimpleNamespace

VALID_FORMULATION = ("laxMPC", "equMPC", "ellipMPC", "MPCT", "HMPC", "ellipHMPC", "personal")
VALID_METHOD = ("ADMM", "SADMM", "EADMM", "FISTA")
VALID_PLATFORM = ("C", "Matlab", "HIP")
VALID_PRECISION = ("double", "float")

ACCEPTED_METHODS = {
    "laxMPC": ("ADMM", "FISTA"), "equMPC": ("ADMM", "FISTA"), "ellipMPC": ("ADMM",),
    "MPCT": ("ADMM", "EADMM"), "HMPC": ("ADMM", "SADMM"), "ellipHMPC": ("ADMM",),
}
ACCEPTED_SUBMETHODS = {
    "laxMPC": {"ADMM": ("",), "FISTA": ("",)}, "equMPC": {"ADMM": ("",), "FISTA": ("",)},
    "ellipMPC": {"ADMM": ("", "soc")}, "MPCT": {"ADMM": ("cs", "semiband"), "EADMM": ("",)},
    "HMPC": {"ADMM": ("cs", "split"), "SADMM": ("split",)}, "ellipHMPC": {"ADMM": ("",)},
}
DEF_METHOD = {"laxMPC": "ADMM", "equMPC": "ADMM", "ellipMPC": "ADMM", "MPCT": "EADMM",
              "HMPC": "ADMM", "ellipHMPC": "ADMM"}
DEF_SUBMETHOD = {
    "laxMPC": {"ADMM": "", "FISTA": ""}, "equMPC": {"ADMM": "", "FISTA": ""},
    "ellipMPC": {"ADMM": ""}, "MPCT": {"ADMM": "cs", "EADMM": ""},
    "HMPC": {"ADMM": "", "SADMM": "split"}, "ellipHMPC": {"ADMM": ""},
}

# Class constants (Spcies_options.m:109-126), with the platform default moved to 'HIP'.
_BASIC = dict(
    verbose=1, save_name="", directory="$SPCIES$", override=True, const_are_static=True,
    precision="double", inf_value=1e6, save=True, debug=True, timing=True,
    in_engineering=False, time_varying=False, force_diagonal=True,
)
_PROPERTIES = ("formulation", "method", "submethod", "platform") + tuple(_BASIC)

# def_options_<formulation>_<method>[_<submethod>] of each plugin
# (formulations/+laxMPC/def_options_laxMPC_ADMM.m:20-23 and siblings).
_DEF_SOLVER = {
    ("laxMPC", "ADMM", ""): dict(rho=1e-2, tol=1e-4, k_max=1000, force_vector_rho=False),
    ("equMPC", "ADMM", ""): dict(rho=1e-2, tol=1e-4, k_max=1000, force_vector_rho=False),
    ("laxMPC", "FISTA", ""): dict(tol=1e-4, k_max=1000),
    ("equMPC", "FISTA", ""): dict(tol=1e-4, k_max=1000),
    # formulations/+MPCT/def_options_MPCT_EADMM.m
    # formulations/+ellipMPC/def_options_ellipMPC_ADMM.m
    ("ellipMPC", "ADMM", ""): dict(rho=1e-2, tol=1e-4, tol_p=1e-4, tol_d=1e-4, k_max=1000, force_vector_rho=False),
    # formulations/+ellipMPC/def_options_ellipMPC_ADMM_soc.m
    ("ellipMPC", "ADMM", "soc"): dict(rho=5, sigma=5, tol_p=1e-4, tol_d=1e-4, k_max=1000),
    # formulations/+HMPC/def_options_HMPC_ADMM.m / def_options_HMPC_SADMM.m
    ("HMPC", "ADMM", ""): dict(rho=1e-2, sigma=1e-2, tol_p=1e-4, tol_d=1e-4, k_max=1000, box_constraints=None,
                               sparse=False, use_soc=False, alpha=0.95),
    ("HMPC", "SADMM", ""): dict(rho=1e-2, sigma=1e-2, tol_p=1e-4, tol_d=1e-4, k_max=1000, box_constraints=None,
                                sparse=False, use_soc=False, alpha=0.95),  # reachable through the C-ABI / benchmarks only
    ("HMPC", "ADMM", "split"): dict(rho=1e-2, sigma=1e-2, tol_p=1e-4, tol_d=1e-4, k_max=1000, box_constraints=None,
                                    sparse=False, use_soc=False, alpha=0.95),
    ("HMPC", "SADMM", "split"): dict(rho=1e-2, sigma=1e-2, tol_p=1e-4, tol_d=1e-4, k_max=1000, box_constraints=None,
                                     sparse=False, use_soc=False, alpha=0.95),
    # formulations/+MPCT/def_options_MPCT_ADMM_cs.m
    ("MPCT", "ADMM", "cs"): dict(rho=1e-2, epsilon_x=1e-6, epsilon_u=1e-6, tol=1e-4, tol_p=1e-4, tol_d=1e-4, k_max=1000,
                                 force_vector_rho=False),
    ("MPCT", "EADMM", ""): dict(rho_base=3, rho_mult=20, epsilon_x=1e-6, epsilon_u=1e-6, tol=1e-4, k_max=1000),
}


def _as_dict(obj):
    if obj is None:
        return {}
    if isinstance(obj, dict):
        return dict(obj)
    return dict(vars(obj))


class SpciesOptions:
    """Options holder; ``SpciesOptions(formulation='laxMPC', method='ADMM', options={...}, ...)``."""

    def __init__(self, **kw):
        kw = {k: v for k, v in kw.items()}
        opts = _as_dict(kw.pop("options", None))
        solver_options = _as_dict(kw.pop("solver_options", None))
        self.solver = {}
        for k, v in _BASIC.items():
            setattr(self, k, copy.copy(v))
        self._formulation = ""
        self._method = ""
        self._submethod = ""
        self._platform = "HIP"

        self.formulation = opts.get("formulation", "") or kw.pop("formulation", "") or ""
        kw.pop("formulation", None)
        if "platform" in opts:
            self.platform = opts["platform"]
        if "platform" in kw:
            self.platform = kw.pop("platform")
        if "type" in kw:  # deprecated alias (Spcies_options.m:200-204)
            warnings.warn("Spcies: 'type' is deprecated, use 'formulation'", DeprecationWarning)
            self.formulation = kw.pop("type")

        if "method" in opts:
            self.method = opts["method"]
        if "method" in kw:
            self.method = kw.pop("method")
        elif not self.method and self.formulation and self.formulation != "personal":
            self.method = DEF_METHOD[self.formulation]

        if "submethod" in opts:
            self.submethod = opts["submethod"]
        if "submethod" in kw:
            self.submethod = kw.pop("submethod")
        elif not self.submethod and self.formulation and self.method and self.formulation != "personal":
            self.submethod = DEF_SUBMETHOD[self.formulation].get(self.method, "")
        if "subclass" in kw:  # deprecated alias (:243-247)
            warnings.warn("Spcies: 'subclass' is deprecated, use 'submethod'", DeprecationWarning)
            self.submethod = kw.pop("subclass")

        self.set_default()
        if opts:
            self.set_opt_from_struct(opts)
        if solver_options:
            warnings.warn("Spcies: 'solver_options' is deprecated, use 'options'", DeprecationWarning)
            self.set_opt_from_struct(solver_options)
        for k, v in kw.items():
            if k not in _PROPERTIES:
                raise TypeError(f"SpciesOptions: unknown name-value argument '{k}'")
            setattr(self, k, v)
        if not self.save_name:
            self.save_name = self.formulation

    # -- validated properties (set.formulation / set.method / set.platform, :276-306)
    @property
    def formulation(self):
        return self._formulation

    @formulation.setter
    def formulation(self, v):
        if v and v not in VALID_FORMULATION:
            raise ValueError(f"Spcies_options: formulation {v} is not supported. Please check valid_formulation")
        self._formulation = v or ""

    @property
    def method(self):
        return self._method

    @method.setter
    def method(self, v):
        if v and v not in VALID_METHOD:
            raise ValueError(f"Spcies_options: method {v} is not supported. Please check valid_method")
        self._method = v or ""

    @property
    def submethod(self):
        return self._submethod

    @submethod.setter
    def submethod(self, v):
        self._submethod = v or ""

    @property
    def platform(self):
        return self._platform

    @platform.setter
    def platform(self, v):
        if v not in VALID_PLATFORM:
            raise ValueError(f"Spcies_options: platform {v} is not supported. Please check valid_platform")
        self._platform = v

    # -- behaviour
    def check_method_selection(self):
        if self.formulation and self.formulation != "personal":
            return self.method in ACCEPTED_METHODS[self.formulation]
        return True

    def check_submethod_selection(self):
        if self.formulation and self.formulation != "personal" and self.method:
            return self.submethod in ACCEPTED_SUBMETHODS[self.formulation].get(self.method, ())
        return True

    def set_default(self):
        """``to_default_from_selection`` (:477-516): load the plugin's ``def_options_*`` into ``.solver``."""
        key = (self.formulation, self.method, self.submethod)
        if key in _DEF_SOLVER:
            self.set_opt_from_struct(_DEF_SOLVER[key])
        elif self.formulation and self.verbose > 0:
            warnings.warn(f"no available def_options_{'_'.join(k for k in key if k)}. Using general default options.")

    def set_opt_from_struct(self, opt, force=True):
        """(:556-603) known property names override the property, anything else lands in ``.solver``."""
        for k, v in _as_dict(opt).items():
            if k in ("formulation", "method", "submethod"):
                continue
            if k in _PROPERTIES:
                setattr(self, k, v)
            elif force or k in self.solver:
                self.solver[k] = v
            elif self.verbose > 0:
                warnings.warn(f"Spcies_options.set_opt_from_struct() could not find option named {k}. Ignoring it.")

    def default_defines(self):
        """``default_defCell`` (:655-673) as a dict of the C ``#define`` switches."""
        d = {}
        if self.debug:
            d["DEBUG"] = 1
        if self.timing:
            d["MEASURE_TIME"] = 1
        d["in_engineering"] = int(bool(self.in_engineering))
        d["TIME_VARYING"] = int(bool(self.time_varying))
        if self.force_diagonal:
            d["IS_DIAG"] = 1
        return d

    def solver_ns(self):
        return SimpleNamespace(**self.solver)
