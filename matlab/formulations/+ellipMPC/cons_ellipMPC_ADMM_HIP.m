%% cons_ellipMPC_ADMM_HIP - HIP platform constructor of the ADMM-based ellipMPC solver with the P-projection onto the
% terminal ellipsoid (sibling of cons_ellipMPC_ADMM_C.m:74-118).  Scalar rho: header rho / rho_i, flag bit0; vector rho (:111-117):
% rho_0, rho, rho_N, rho_i_0, rho_i, rho_i_N -> ids 17, 61, 62, 64, 63, 65 (the same ids as cons_laxMPC_ADMM_HIP.m)
function constructor = cons_ellipMPC_ADMM_HIP(recipe)
    vars = ellipMPC.compute_ellipMPC_ADMM_ingredients(recipe.controller, recipe.options);
    hdr = struct('formulation', 4, 'method', 1, 'submethod', 0, 'flags', 0, 'rho', 0, 'rho_i', 0, 'reserved', [0 0 0 0 vars.r]);
    arrays = {1, vars.AB, false; 2, vars.Alpha, false; 3, vars.Beta, false; 4, vars.Hi, false; 5, vars.Hi_0(:), false; ...
              6, vars.Hi_N, false; 7, vars.Q(:), false; 8, vars.R(:), false; 9, vars.T, false; 53, vars.P, false; ...
              54, vars.P_half, false; 55, vars.Pinv_half, false; 56, vars.c(:), false; 57, vars.LBz, false; 58, vars.UBz, false; ...
              59, vars.LBu0(:), false; 60, vars.UBu0(:), false};
    if vars.rho_is_scalar
        hdr.flags = 1; hdr.rho = vars.rho; hdr.rho_i = vars.rho_i;
    else
        arrays = [arrays; {17, vars.rho_0(:), false; 61, vars.rho, false; 62, vars.rho_N(:), false; ...
                           64, vars.rho_i_0(:), false; 63, vars.rho_i, false; 65, vars.rho_i_N(:), false}];
    end
    constructor = HIP.cons_generic(recipe, vars, hdr, arrays, 0, 'ellipMPC');
end
