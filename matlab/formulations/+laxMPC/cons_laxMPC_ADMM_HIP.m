%% cons_laxMPC_ADMM_HIP
%
% Constructor of the HIP (AMD MI355X) platform for the ADMM-based laxMPC solver.
% Drop-in sibling of cons_laxMPC_ADMM_C.m / cons_laxMPC_ADMM_Matlab.m: spcies_gen_controller
% reaches it through the usual name-mangled dispatch  cons_<formulation>_<method>_<platform>
% once 'HIP' is listed in Spcies_options.valid_platform.
%
% It reuses the toolbox's own offline computation (laxMPC.compute_laxMPC_ADMM_ingredients), writes
% the ingredients as a binary problem blob (layout: include/spcies_hip.h) and asks the constructor
% to build the mex gateway struct_laxMPC_ADMM_HIP_Matlab.c against libspcies_hip.so.
%
% INPUTS / OUTPUTS: as cons_laxMPC_ADMM_C.m.

function constructor = cons_laxMPC_ADMM_HIP(recipe)

    import sp_utils.add_line

    full_path = mfilename('fullpath');
    this_path = fileparts(full_path);

    %% Ingredients: exactly what the C platform prints as constants
    vars = laxMPC.compute_laxMPC_ADMM_ingredients(recipe.controller, recipe.options);
    if ~vars.rho_is_scalar
        error('Spcies:laxMPC:HIP:vector_rho', 'The HIP platform currently supports scalar rho only');
    end
    if recipe.options.time_varying
        error('Spcies:laxMPC:HIP:time_varying', 'The HIP platform does not support time_varying yet');
    end
    n = vars.n; m = vars.m; N = vars.N;

    %% Write the problem blob
    save_dir = recipe.options.directory;
    if strcmp(save_dir, '$SPCIES$'); save_dir = [spcies_get_root_directory '/generated_solvers/']; end
    blob_path = [save_dir recipe.options.save_name '.spcb'];
    HIP.write_blob(blob_path, 1, 1, vars, recipe.options.solver.k_max, recipe.options.solver.tol);

    %% Defines consumed by the mex gateway
    defCell = recipe.options.default_defCell();
    defCell = add_line(defCell, 'nn_', n, 1, 'uint', 'define');
    defCell = add_line(defCell, 'mm_', m, 1, 'uint', 'define');
    defCell = add_line(defCell, 'nm_', n+m, 1, 'uint', 'define');
    defCell = add_line(defCell, 'NN_', N, 1, 'uint', 'define');
    defCell = add_line(defCell, 'dim_', N*(n+m), 1, 'uint', 'define');

    %% Constructor
    constructor = Spcies_constructor;
    constructor = constructor.new_empty_file('mex_code', recipe.options, 'c');
    constructor.files.mex_code.blocks = {'$START$', [this_path '/struct_laxMPC_ADMM_HIP_Matlab.c']};
    constructor.files.mex_code.flags = {'$FORM$', 'laxMPC'; 'BLOB_PATH', ['"' blob_path '"']};
    constructor.files.mex_code.exec_me = ['mex -silent $INSERT_PATH$$INSERT_NAME$.c -outdir $INSERT_PATH$ ' ...
        '-I' HIP.engine_root() '/include -L' HIP.engine_root() '/spcies_amd -lspcies_hip'];
    constructor.data = {'$INSERT_DEFINES$', defCell};

end
