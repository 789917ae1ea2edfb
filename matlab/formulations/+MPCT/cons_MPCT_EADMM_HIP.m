%% cons_MPCT_EADMM_HIP - HIP platform constructor of the EADMM-based MPCT solver (sibling of cons_MPCT_EADMM_C.m:82-108)
% Diagonal Q, R (vars.H3i exists, IS_DIAG == 1): H3i travels (id 25).  General Q, R (:99-107): the six dense blocks
% Q_base_inv, Q_mult_inv, R_base_inv, R_mult_inv, AB_base_inv, AB_mult_inv travel instead (ids 91-96), header flag bit5.
function constructor = cons_MPCT_EADMM_HIP(recipe)
    vars = MPCT.compute_MPCT_EADMM_ingredients(recipe.controller, recipe.options);
    hdr = struct('formulation', 3, 'method', 3, 'submethod', 0, 'flags', 1);
    arrays = {1, vars.AB, false; 2, vars.Alpha, false; 3, vars.Beta, false; 10, vars.LB(:), false; 11, vars.UB(:), false; ...
              9, vars.T, false; 15, vars.S, false; 16, vars.rho, false; 17, vars.rho_0(:), false; 18, vars.rho_s(:), false; ...
              19, vars.LB0(:), false; 20, vars.UB0(:), false; 21, vars.LBs(:), false; 22, vars.UBs(:), false; ...
              23, vars.H1i, false; 24, vars.W2, false};
    if isfield(vars, 'H3i')
        arrays = [arrays; {25, vars.H3i, false}];
    else
        hdr.flags = bitor(hdr.flags, 32);
        arrays = [arrays; {91, vars.Q_base_inv, false; 92, vars.Q_mult_inv, false; 93, vars.R_base_inv, false; ...
                           94, vars.R_mult_inv, false; 95, vars.AB_base_inv, false; 96, vars.AB_mult_inv, false}];
    end
    constructor = HIP.cons_generic(recipe, vars, hdr, arrays, 0, 'MPCT');
end
