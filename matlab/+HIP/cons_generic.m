%% HIP.cons_generic - shared body of the cons_<formulation>_<method>[_<submethod>]_HIP constructors
%
% Drop-in sibling of the toolbox's cons_*_C.m files: the caller computes `vars` with the toolbox's own
% compute_*_ingredients function and lists which of them travel (ids of include/spcies_hip.h); this function
% writes the blob and asks the constructor to build the generic mex gateway against libspcies_hip.so.
%   hdr     - header struct for HIP.write_blob_generic (formulation, method, submethod, flags, reserved; the rest
%             is filled here from vars / options)
%   arrays  - cell {id, value, is_int}
%   n_extra - extra inputs of the generated function: 0, 1 (ellipMPC soc: r) or 6 (time-varying: A, B, Q, R, LB, UB)
function constructor = cons_generic(recipe, vars, hdr, arrays, n_extra, form_name)
    import sp_utils.add_line
    hdr.n = vars.n; hdr.m = vars.m; hdr.N = vars.N;
    hdr.k_max = recipe.options.solver.k_max;
    if isfield(recipe.options.solver, 'tol'); hdr.tol = recipe.options.solver.tol; else; hdr.tol = recipe.options.solver.tol_p; end
    if ~isfield(hdr, 'rho'); hdr.rho = 0; hdr.rho_i = 0; end
    if ~isfield(hdr, 'reserved'); hdr.reserved = zeros(1, 5); end
    if recipe.options.in_engineering   % code_laxMPC_ADMM_C.c:83-100, 642-646
        hdr.flags = bitor(hdr.flags, 8);
        arrays = [arrays; {48, vars.scaling_x(:), false; 49, vars.scaling_u(:), false; 50, vars.scaling_i_u(:), false; ...
                           51, vars.OpPoint_x(:), false; 52, vars.OpPoint_u(:), false}];
    end
    save_dir = recipe.options.directory;
    if strcmp(save_dir, '$SPCIES$'); save_dir = [spcies_get_root_directory '/generated_solvers/']; end
    blob_path = [save_dir recipe.options.save_name '.spcb'];
    HIP.write_blob_generic(blob_path, hdr, arrays);

    defCell = recipe.options.default_defCell();
    defCell = add_line(defCell, 'nn_', vars.n, 1, 'uint', 'define');
    defCell = add_line(defCell, 'mm_', vars.m, 1, 'uint', 'define');
    defCell = add_line(defCell, 'N_EXTRA_', n_extra, 1, 'uint', 'define');

    this_path = fileparts(mfilename('fullpath'));
    constructor = Spcies_constructor;
    constructor = constructor.new_empty_file('mex_code', recipe.options, 'c');
    constructor.files.mex_code.blocks = {'$START$', [this_path '/struct_generic_HIP_Matlab.c']};
    constructor.files.mex_code.flags = {'$FORM$', form_name; 'BLOB_PATH', ['"' blob_path '"']};
    constructor.files.mex_code.exec_me = ['mex -silent $INSERT_PATH$$INSERT_NAME$.c -outdir $INSERT_PATH$ ' ...
        '-I' HIP.engine_root() '/include -L' HIP.engine_root() '/spcies_amd -lspcies_hip'];
    constructor.data = {'$INSERT_DEFINES$', defCell};
end
