"""``spcies_gen_controller`` of the HIP platform.

Same name-value interface as the reference's ``spcies_gen_controller.m:72-133`` (``sys``, ``param``,
``options``, ``platform``, ``formulation``, ``method``, ``submethod`` and the deprecated ``type`` /
``solver_options`` the reference's own tests still use, ``tests/test_laxMPC_ADMM.m:24-25``), same
name-mangled dispatch ``cons_<formulation>_<method>[_<submethod>]_<platform>`` (``:114-130``).
Where the reference writes ``<name>.c/.h`` + a mex, the HIP constructor returns the callable
:class:`~spcies_amd.solver.HipSolver` (``u, k, e_flag, sol = solver(x0, xr, ur)``).
"""
from __future__ import annotations

from types import SimpleNamespace

from .formulations import HMPC as _hmpc
from .formulations import MPCT as _mpct
from .formulations import ellipMPC as _ellip
from .formulations import laxMPC as _lax
from .options import SpciesOptions
from .solver import HipSolver


def _eng(v, recipe):
    """Option ``in_engineering``: scaling vectors and operating point travel with the controller constants."""
    return _lax.add_engineering(v, _lax._get(recipe.controller, "sys"), recipe.options)


def cons_laxMPC_ADMM_HIP(recipe, device=0):
    v = _eng(_lax.compute_laxMPC_ADMM_ingredients(recipe.controller, recipe.options), recipe)
    return HipSolver(v, device=device, name=recipe.options.save_name, debug=recipe.options.debug)


def cons_equMPC_ADMM_HIP(recipe, device=0):
    v = _eng(_lax.compute_equMPC_ADMM_ingredients(recipe.controller, recipe.options), recipe)
    return HipSolver(v, device=device, name=recipe.options.save_name, debug=recipe.options.debug)


def cons_laxMPC_FISTA_HIP(recipe, device=0):
    v = _eng(_lax.compute_laxMPC_FISTA_ingredients(recipe.controller, recipe.options), recipe)
    return HipSolver(v, device=device, name=recipe.options.save_name, debug=recipe.options.debug)


def cons_equMPC_FISTA_HIP(recipe, device=0):
    v = _eng(_lax.compute_equMPC_FISTA_ingredients(recipe.controller, recipe.options), recipe)
    return HipSolver(v, device=device, name=recipe.options.save_name, debug=recipe.options.debug)


def cons_MPCT_EADMM_HIP(recipe, device=0):
    v = _eng(_mpct.compute_MPCT_EADMM_ingredients(recipe.controller, recipe.options), recipe)
    return HipSolver(v, device=device, name=recipe.options.save_name, debug=recipe.options.debug)


def cons_ellipMPC_ADMM_HIP(recipe, device=0):
    v = _eng(_ellip.compute_ellipMPC_ADMM_ingredients(recipe.controller, recipe.options), recipe)
    return HipSolver(v, device=device, name=recipe.options.save_name, debug=recipe.options.debug)


def cons_ellipMPC_ADMM_soc_HIP(recipe, device=0):
    v = _eng(_ellip.compute_ellipMPC_ADMM_soc_ingredients(recipe.controller, recipe.options), recipe)
    return HipSolver(v, device=device, name=recipe.options.save_name, debug=recipe.options.debug)


def cons_HMPC_ADMM_split_HIP(recipe, device=0):
    v = _eng(_hmpc.compute_HMPC_ADMM_split_ingredients(recipe.controller, recipe.options), recipe)
    return HipSolver(v, device=device, name=recipe.options.save_name, debug=recipe.options.debug)


def cons_MPCT_ADMM_cs_HIP(recipe, device=0):
    """MPCT ADMM on the extended state space (cons_MPCT_ADMM_cs_C.m:40-131)."""
    v = _eng(_mpct.compute_MPCT_ADMM_cs_ingredients(recipe.controller, recipe.options), recipe)
    return HipSolver(v, device=device, name=recipe.options.save_name, debug=recipe.options.debug)


def cons_HMPC_ADMM_HIP(recipe, device=0):
    """HMPC ADMM without the splitting - the reference's default HMPC solver (cons_HMPC_ADMM_C.m:47-151)."""
    v = _eng(_hmpc.compute_HMPC_ADMM_ingredients(recipe.controller, recipe.options), recipe)
    return HipSolver(v, device=device, name=recipe.options.save_name, debug=recipe.options.debug)


cons_HMPC_SADMM_split_HIP = cons_HMPC_ADMM_split_HIP  # same ingredients; the method selects the symmetric dual steps

_CONSTRUCTORS = {f.__name__: f for f in (cons_laxMPC_ADMM_HIP, cons_equMPC_ADMM_HIP, cons_laxMPC_FISTA_HIP,
                                         cons_equMPC_FISTA_HIP, cons_MPCT_EADMM_HIP, cons_ellipMPC_ADMM_HIP,
                                         cons_ellipMPC_ADMM_soc_HIP,
                                         cons_HMPC_ADMM_split_HIP, cons_HMPC_ADMM_HIP, cons_MPCT_ADMM_cs_HIP)}
_CONSTRUCTORS["cons_HMPC_SADMM_split_HIP"] = cons_HMPC_SADMM_split_HIP


def spcies_gen_controller(*, sys=None, param=None, device=0, **kw):
    if sys is None:
        raise ValueError("spcies_gen_controller: a 'sys' structure must be provided")
    if param is None:
        raise ValueError("spcies_gen_controller: a 'param' structure must be provided")
    options = SpciesOptions(**kw)
    if not options.formulation:
        raise ValueError("Spcies:input_error:no_formulation - the formulation field of options is empty")
    if not options.check_method_selection() or not options.check_submethod_selection():
        raise ValueError(f"Spcies: method '{options.method}'/submethod '{options.submethod}' not accepted for "
                         f"formulation '{options.formulation}'")
    recipe = SimpleNamespace(controller=SimpleNamespace(sys=sys, param=param), options=options)
    cons_name = "cons_" + options.formulation
    if options.method:
        cons_name += "_" + options.method
    if options.submethod:
        cons_name += "_" + options.submethod
    cons_name += "_" + options.platform
    if options.platform != "HIP":
        raise NotImplementedError(f"{cons_name}: this package only builds the 'HIP' platform; the 'C' and 'Matlab' "
                                  "platforms are the reference toolbox's own")
    if cons_name not in _CONSTRUCTORS:
        raise NotImplementedError(f"{cons_name} is not built yet on the HIP platform")
    return _CONSTRUCTORS[cons_name](recipe, device=device)
