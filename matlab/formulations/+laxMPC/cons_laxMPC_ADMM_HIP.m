%% cons_laxMPC_ADMM_HIP - HIP (AMD MI355X) platform constructor of the ADMM-based laxMPC solver
%
% Drop-in sibling of cons_laxMPC_ADMM_C.m / cons_laxMPC_ADMM_Matlab.m: spcies_gen_controller reaches it through the usual
% name-mangled dispatch cons_<formulation>_<method>_<platform> once 'HIP' is listed in Spcies_options.valid_platform.
% It reuses the toolbox's own offline computation (laxMPC.compute_laxMPC_ADMM_ingredients) and ships what
% cons_laxMPC_ADMM_C.m:82-130 prints as constants, switch by switch:
%   scalar rho (SCALAR_RHO, :119-122)           header rho / rho_i, flag bit0
%   vector rho (:123-129)                       rho_0, rho, rho_N, rho_i_0, rho_i, rho_i_N -> ids 17, 61, 62, 64, 63, 65
%   one bound column per prediction step (VAR_BOUNDS, :82-90)   LB0 / UB0, LB / UB, LBN / UBN -> ids 19, 20, 10, 11, 66, 67; flag bit4
%   time_varying (:92-109)                      only T and T_rho_i travel (ids 9, 47), flag bit2; the generated function then
%                                               takes A, B, Q, R, LB, UB with every call (struct_laxMPC_ADMM_C_Matlab.c:29-31)
%   in_engineering (:110-116)                   added by HIP.cons_generic (ids 48-52, flag bit3)
function constructor = cons_laxMPC_ADMM_HIP(recipe)
    vars = laxMPC.compute_laxMPC_ADMM_ingredients(recipe.controller, recipe.options);
    n = vars.n;
    hdr = struct('formulation', 1, 'method', 1, 'submethod', 0, 'flags', 0, 'rho', 0, 'rho_i', 0);
    if recipe.options.time_varying
        if ~vars.rho_is_scalar
            error('Spcies:laxMPC:HIP:time_varying', 'time_varying solvers take a scalar rho (cons_laxMPC_ADMM_C.m:50)');
        end
        hdr.flags = 1 + 4; hdr.rho = vars.rho; hdr.rho_i = vars.rho_i;
        arrays = {9, vars.T, false; 47, vars.T_rho_i, false};
        constructor = HIP.cons_generic(recipe, vars, hdr, arrays, 6, 'laxMPC');
        return
    end
    arrays = {1, vars.AB, false; 2, vars.Alpha, false; 3, vars.Beta, false; 4, vars.Hi, false; 5, vars.Hi_0(:), false; ...
              6, vars.Hi_N, false; 7, vars.Q(:), false; 8, vars.R(:), false; 9, vars.T, false};
    if size(vars.LB, 2) > 1
        hdr.flags = bitor(hdr.flags, 16);
        arrays = [arrays; {19, vars.LB(n+1:end, 1), false; 20, vars.UB(n+1:end, 1), false; 10, vars.LB(:, 2:end-1)', false; ...
                           11, vars.UB(:, 2:end-1)', false; 66, vars.LB(1:n, end), false; 67, vars.UB(1:n, end), false}];
    else
        arrays = [arrays; {10, vars.LB(:), false; 11, vars.UB(:), false}];
    end
    if vars.rho_is_scalar
        hdr.flags = bitor(hdr.flags, 1); hdr.rho = vars.rho; hdr.rho_i = vars.rho_i;
    else
        arrays = [arrays; {17, vars.rho_0(:), false; 61, vars.rho, false; 62, vars.rho_N(:), false; ...
                           64, vars.rho_i_0(:), false; 63, vars.rho_i, false; 65, vars.rho_i_N(:), false}];
    end
    constructor = HIP.cons_generic(recipe, vars, hdr, arrays, 0, 'laxMPC');
end
