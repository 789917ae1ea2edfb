%% cons_laxMPC_FISTA_HIP - HIP platform constructor of the FISTA-based laxMPC solver (sibling of cons_laxMPC_FISTA_C.m:94-107)
function constructor = cons_laxMPC_FISTA_HIP(recipe)
    vars = laxMPC.compute_laxMPC_FISTA_ingredients(recipe.controller, recipe.options);
    hdr = struct('formulation', 1, 'method', 2, 'submethod', 0, 'flags', 1);
    arrays = {1, vars.AB, false; 2, vars.Alpha, false; 3, vars.Beta, false; 7, vars.Q(:), false; 8, vars.R(:), false; ...
              12, vars.QRi(:), false; 13, vars.T(:), false; 14, vars.Ti(:), false; 10, vars.LB(:), false; 11, vars.UB(:), false};
    constructor = HIP.cons_generic(recipe, vars, hdr, arrays, 0, 'laxMPC');
end
