%% cons_HMPC_ADMM_HIP - HIP platform constructor of the HMPC ADMM / SADMM solver WITHOUT the splitting
% (sibling of cons_HMPC_ADMM_C.m:47-151 - the reference's default HMPC solver; box or coupled constraints: LBy / UBy carry
% n_y entries and LB / UB N n_y with coupled constraints, the engine reads the mode from those counts).  The dense M1, M2 and
% the CSR forms of C and C' travel as the generator prints them (:113-131, 0-based indices); the engine runs the
% z-update as one dgemm per iteration for the whole batch.  Solution record: z, s, lambda (header_HMPC_ADMM_C.h:14-22).
function constructor = cons_HMPC_ADMM_HIP(recipe)
    if isempty(recipe.options.solver.box_constraints)
        recipe.options.solver.box_constraints = ~isfield(recipe.controller.sys, 'E');  % cons_HMPC_ADMM_C.m:58-64
    end
    vars = HMPC.compute_HMPC_ADMM_ingredients(recipe.controller, recipe.options);
    o = recipe.options.solver;
    is_sadmm = strcmp(recipe.options.method, 'SADMM');
    alpha = 0; if is_sadmm; alpha = o.alpha; end
    hdr = struct('formulation', 5, 'method', 1 + 3*is_sadmm, 'submethod', 0, 'flags', 1 + 2*o.use_soc, 'rho', vars.rho, ...
                 'rho_i', vars.rho_i, 'reserved', [0 0 o.tol_d alpha 0]);
    arrays = {26, vars.A, false; 7, vars.Q, false; 41, vars.Te, false; 42, vars.Se, false; 10, vars.LB(:), false; ...
              11, vars.UB(:), false; 43, vars.LBy(:), false; 44, vars.UBy(:), false; ...
              71, vars.C_CSR.val(:), false; 72, vars.C_CSR.col(:) - 1, true; 73, vars.C_CSR.row(:) - 1, true; ...
              74, vars.Ct_CSR.val(:), false; 75, vars.Ct_CSR.col(:) - 1, true; 76, vars.Ct_CSR.row(:) - 1, true; ...
              68, vars.M1, false; 69, vars.M2, false};
    if o.use_soc
        arrays = [arrays; {77, vars.d(:), false}];
    end
    constructor = HIP.cons_generic(recipe, vars, hdr, arrays, 0, 'HMPC');
end
