%% cons_HMPC_ADMM_split_HIP - HIP platform constructor of the HMPC ADMM / SADMM solver with the (z_hat, s_hat) splitting
% (sibling of cons_HMPC_ADMM_split_C.m:88-181 and cons_HMPC_SADMM_split_C.m; box constraints, or coupled output constraints
% LBy <= E x + F u <= UBy - header flag bit6, s then starts with the N n_y box slacks of the outputs).  Both ways of solving
% the KKT system travel: the L D L' factor (sparse path; idx_x0 0-based inside bh as :140) and the dense M1, M2 of the
% reference's default NON_SPARSE path, which the engine runs as one dgemm per iteration.
function constructor = cons_HMPC_ADMM_split_HIP(recipe)
    if isempty(recipe.options.solver.box_constraints)
        recipe.options.solver.box_constraints = ~isfield(recipe.controller.sys, 'E');  % cons_HMPC_ADMM_split_C.m:51-56
    end
    vars = HMPC.compute_HMPC_ADMM_split_ingredients(recipe.controller, recipe.options);
    o = recipe.options.solver;
    is_sadmm = strcmp(recipe.options.method, 'SADMM');
    alpha = 0; if is_sadmm; alpha = o.alpha; end
    hdr = struct('formulation', 5, 'method', 1 + 3*is_sadmm, 'submethod', 2, 'flags', 1 + 2*o.use_soc + 64*(~o.box_constraints), ...
                 'rho', vars.rho, 'rho_i', vars.rho_i, 'reserved', [vars.sigma vars.sigma_i o.tol_d alpha 0]);
    dim = vars.dim; n_s = vars.n_s;
    bh_natural = [vars.b(:); vars.d(:)];                              % compute_HMPC_ADMM_split_ingredients.m:223
    perm_bh = vars.Pldl' * [zeros(dim + n_s, 1); bh_natural];        % :276-279 (what `var.bh` holds when sparse = true)
    arrays = {26, vars.A, false; 7, vars.Q, false; 41, vars.Te, false; 42, vars.Se, false; 10, vars.LB(:), false; ...
              11, vars.UB(:), false; 43, vars.LBy(:), false; 44, vars.UBy(:), false; 28, vars.L_CSC.val(:), false; ...
              29, vars.L_CSC.col(:) - 1, true; 30, vars.L_CSC.row(:) - 1, true; 31, vars.Dinv(:), false; ...
              45, vars.idx_x0(:) - 1 - dim - n_s, true; 46, perm_bh(dim + n_s + 1:end), false; ...
              68, vars.M1, false; 69, vars.M2, false; 70, bh_natural, false};
    constructor = HIP.cons_generic(recipe, vars, hdr, arrays, 0, 'HMPC');
end
