%% cons_HMPC_SADMM_split_HIP - the symmetric-ADMM solver shares the ADMM constructor (cons_HMPC_SADMM_split_C.m:40-46)
function constructor = cons_HMPC_SADMM_split_HIP(recipe)
    constructor = HMPC.cons_HMPC_ADMM_split_HIP(recipe);
end
