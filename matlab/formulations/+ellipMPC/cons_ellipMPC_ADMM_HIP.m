%% cons_ellipMPC_ADMM_HIP - HIP platform constructor of the ADMM-based ellipMPC solver with the P-projection onto the
% terminal ellipsoid (sibling of cons_ellipMPC_ADMM_C.m:74-110; scalar rho)
function constructor = cons_ellipMPC_ADMM_HIP(recipe)
    vars = ellipMPC.compute_ellipMPC_ADMM_ingredients(recipe.controller, recipe.options);
    if ~vars.rho_is_scalar
        error('Spcies:ellipMPC:HIP:vector_rho', 'The HIP platform supports scalar rho only for ellipMPC');
    end
    hdr = struct('formulation', 4, 'method', 1, 'submethod', 0, 'flags', 1, 'rho', vars.rho, 'rho_i', vars.rho_i, ...
                 'reserved', [0 0 0 0 vars.r]);
    arrays = {1, vars.AB, false; 2, vars.Alpha, false; 3, vars.Beta, false; 4, vars.Hi, false; 5, vars.Hi_0(:), false; ...
              6, vars.Hi_N, false; 7, vars.Q(:), false; 8, vars.R(:), false; 9, vars.T, false; 53, vars.P, false; ...
              54, vars.P_half, false; 55, vars.Pinv_half, false; 56, vars.c(:), false; 57, vars.LBz, false; 58, vars.UBz, false; ...
              59, vars.LBu0(:), false; 60, vars.UBu0(:), false};
    constructor = HIP.cons_generic(recipe, vars, hdr, arrays, 0, 'ellipMPC');
end
