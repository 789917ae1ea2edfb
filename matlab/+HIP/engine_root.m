%% HIP.engine_root - directory of the spcies-hip engine checkout (holds include/ and spcies_amd/)
function p = engine_root()
    p = getenv('SPCIES_HIP_ROOT');
    if isempty(p)
        error('Spcies:HIP:root', 'Set the environment variable SPCIES_HIP_ROOT to the spcies-hip checkout');
    end
end
