%% cons_equMPC_FISTA_HIP - HIP platform constructor of the FISTA-based equMPC solver (sibling of cons_equMPC_FISTA_C.m)
function constructor = cons_equMPC_FISTA_HIP(recipe)
    vars = equMPC.compute_equMPC_FISTA_ingredients(recipe.controller, recipe.options);
    n = vars.n;
    hdr = struct('formulation', 2, 'method', 2, 'submethod', 0, 'flags', 1);
    arrays = {1, vars.AB, false; 2, vars.Alpha, false; 3, vars.Beta, false; 7, vars.Q(:), false; 8, vars.R(:), false; ...
              12, vars.QRi(:), false; 13, zeros(n, 1), false; 14, zeros(n, 1), false; 10, vars.LB(:), false; 11, vars.UB(:), false};
    constructor = HIP.cons_generic(recipe, vars, hdr, arrays, 0, 'equMPC');
end
