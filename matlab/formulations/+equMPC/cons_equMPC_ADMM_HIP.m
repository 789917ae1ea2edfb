%% cons_equMPC_ADMM_HIP - HIP platform constructor of the ADMM-based equMPC solver (sibling of cons_equMPC_ADMM_C.m)
% Scalar rho, constant bounds; with recipe.options.time_varying only T and T_rho_i travel (flag bit2) and the
% generated function takes the model with every call (struct_laxMPC_ADMM_C_Matlab.c:29-31).
function constructor = cons_equMPC_ADMM_HIP(recipe)
    vars = equMPC.compute_equMPC_ADMM_ingredients(recipe.controller, recipe.options);
    n = vars.n;
    hdr = struct('formulation', 2, 'method', 1, 'submethod', 0, 'flags', 1, 'rho', vars.rho, 'rho_i', vars.rho_i);
    if recipe.options.time_varying
        hdr.flags = bitor(hdr.flags, 4);
        arrays = {9, zeros(n), false; 47, zeros(n), false};
        constructor = HIP.cons_generic(recipe, vars, hdr, arrays, 6, 'equMPC');
        return
    end
    arrays = {1, vars.AB, false; 2, vars.Alpha, false; 3, vars.Beta, false; 4, vars.Hi, false; 5, vars.Hi_0(:), false; ...
              6, zeros(n), false; 7, vars.Q(:), false; 8, vars.R(:), false; 9, zeros(n), false; ...
              10, vars.LB(:), false; 11, vars.UB(:), false};
    constructor = HIP.cons_generic(recipe, vars, hdr, arrays, 0, 'equMPC');
end
