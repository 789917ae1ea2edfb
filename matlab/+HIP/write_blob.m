%% HIP.write_blob - serialise solver ingredients into the engine's problem blob
%
% Layout: include/spcies_hip.h (128-byte header, 48-byte directory entries, 64-byte aligned FP64
% payloads, little endian).  MATLAB arrays are column-major; the engine wants the reference's
% C order ([row][col], 3-D as [k][i][j] - what dec_var.m prints), hence the permutes.
%
% formulation: 1 laxMPC, 2 equMPC.   method: 1 ADMM.

function write_blob(path, formulation, method, vars, k_max, tol)
    n = vars.n; m = vars.m; N = vars.N;
    if formulation == 2
        vars.Hi_N = zeros(n); vars.T = zeros(n);
    end
    inf_value = 1e20;
    LB = max(min(vars.LB(:), inf_value), -inf_value);
    UB = max(min(vars.UB(:), inf_value), -inf_value);
    arrays = { 1, vars.AB.';  2, permute(vars.Alpha, [2 1 3]);  3, permute(vars.Beta, [2 1 3]); ...
               4, vars.Hi.';  5, vars.Hi_0(:);  6, vars.Hi_N.';  7, vars.Q(:);  8, vars.R(:); ...
               9, vars.T.';  10, LB;  11, UB };
    dims = { [n n+m 0 0]; [N-1 n n 0]; [N n n 0]; [N-1 n+m 0 0]; [m 0 0 0]; [n n 0 0]; [n 0 0 0]; ...
             [m 0 0 0]; [n n 0 0]; [n+m 0 0 0]; [n+m 0 0 0] };
    na = size(arrays, 1);
    align = @(x) ceil(x/64)*64;
    off = align(128 + 48*na);
    offs = zeros(na, 1);
    for i = 1:na
        offs(i) = off;
        off = align(off + 8*numel(arrays{i, 2}));
    end
    total = off;
    f = fopen(path, 'w', 'ieee-le');
    fwrite(f, 'SPCSBLB1', 'char');
    fwrite(f, [1 128 formulation method 0 1 n m N k_max na 0], 'uint32');
    fwrite(f, total, 'uint64');
    fwrite(f, [tol vars.rho vars.rho_i 0 0 0 0 0], 'double');
    for i = 1:na
        fwrite(f, [arrays{i, 1} 0], 'uint32');
        fwrite(f, [offs(i) numel(arrays{i, 2})], 'uint64');
        fwrite(f, [dims{i} 0 0], 'uint32');
    end
    for i = 1:na
        fwrite(f, zeros(offs(i) - ftell(f), 1), 'uint8');
        fwrite(f, arrays{i, 2}(:), 'double');
    end
    fwrite(f, zeros(total - ftell(f), 1), 'uint8');
    fclose(f);
end
