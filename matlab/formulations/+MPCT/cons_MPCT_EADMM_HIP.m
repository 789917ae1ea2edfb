%% cons_MPCT_EADMM_HIP - HIP platform constructor of the EADMM-based MPCT solver (sibling of cons_MPCT_EADMM_C.m:82-100,
% diagonal Q, R path: H3i)
function constructor = cons_MPCT_EADMM_HIP(recipe)
    vars = MPCT.compute_MPCT_EADMM_ingredients(recipe.controller, recipe.options);
    hdr = struct('formulation', 3, 'method', 3, 'submethod', 0, 'flags', 1);
    arrays = {1, vars.AB, false; 2, vars.Alpha, false; 3, vars.Beta, false; 10, vars.LB(:), false; 11, vars.UB(:), false; ...
              9, vars.T, false; 15, vars.S, false; 16, vars.rho, false; 17, vars.rho_0(:), false; 18, vars.rho_s(:), false; ...
              19, vars.LB_0(:), false; 20, vars.UB_0(:), false; 21, vars.LB_s(:), false; 22, vars.UB_s(:), false; ...
              23, vars.H1i, false; 24, vars.W2, false; 25, vars.H3i, false};
    constructor = HIP.cons_generic(recipe, vars, hdr, arrays, 0, 'MPCT');
end
