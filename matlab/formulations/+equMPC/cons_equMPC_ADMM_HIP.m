%% cons_equMPC_ADMM_HIP - HIP platform constructor of the ADMM-based equMPC solver (sibling of cons_equMPC_ADMM_C.m)
% Same switches as cons_laxMPC_ADMM_HIP.m (scalar / vector rho, VAR_BOUNDS, time_varying, in_engineering); no terminal block:
% Hi_N and T travel as zeros, rho_N / rho_i_N and LBN / UBN do not exist.
function constructor = cons_equMPC_ADMM_HIP(recipe)
    vars = equMPC.compute_equMPC_ADMM_ingredients(recipe.controller, recipe.options);
    n = vars.n;
    hdr = struct('formulation', 2, 'method', 1, 'submethod', 0, 'flags', 0, 'rho', 0, 'rho_i', 0);
    if recipe.options.time_varying
        hdr.flags = 1 + 4; hdr.rho = vars.rho; hdr.rho_i = vars.rho_i;
        arrays = {9, zeros(n), false; 47, zeros(n), false};
        constructor = HIP.cons_generic(recipe, vars, hdr, arrays, 6, 'equMPC');
        return
    end
    arrays = {1, vars.AB, false; 2, vars.Alpha, false; 3, vars.Beta, false; 4, vars.Hi, false; 5, vars.Hi_0(:), false; ...
              6, zeros(n), false; 7, vars.Q(:), false; 8, vars.R(:), false; 9, zeros(n), false};
    if size(vars.LB, 2) > 1
        hdr.flags = bitor(hdr.flags, 16);
        arrays = [arrays; {19, vars.LB(n+1:end, 1), false; 20, vars.UB(n+1:end, 1), false; 10, vars.LB(:, 2:end-1)', false; ...
                           11, vars.UB(:, 2:end-1)', false}];
    else
        arrays = [arrays; {10, vars.LB(:), false; 11, vars.UB(:), false}];
    end
    if vars.rho_is_scalar
        hdr.flags = bitor(hdr.flags, 1); hdr.rho = vars.rho; hdr.rho_i = vars.rho_i;
    else
        arrays = [arrays; {17, vars.rho_0(:), false; 61, vars.rho, false; 64, vars.rho_i_0(:), false; 63, vars.rho_i, false}];
    end
    constructor = HIP.cons_generic(recipe, vars, hdr, arrays, 0, 'equMPC');
end
