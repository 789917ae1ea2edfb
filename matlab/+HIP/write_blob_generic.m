%% HIP.write_blob_generic - serialise any solver's ingredients into the engine's problem blob
%
% Layout: include/spcies_hip.h (128-byte header, 48-byte directory entries, 64-byte aligned payloads, little
% endian).  hdr: struct with fields formulation, method, submethod, flags, n, m, N, k_max, tol, rho, rho_i and
% reserved (1 x 5: sigma, 1/sigma, tol_d, alpha, r).  arrays: cell {id, value, is_int}; values are given as MATLAB
% arrays and written in the reference's C order ([row][col]; 3-D as [k][i][j], what dec_var.m prints): 2-D arrays
% are transposed, 3-D ones permuted [2 1 3]; index arrays must already be 0-based.  +-Inf bounds become +-1e20.
function write_blob_generic(path, hdr, arrays)
    na = size(arrays, 1);
    align = @(x) ceil(x/64)*64;
    payload = cell(na, 1); dims = zeros(na, 4);
    for i = 1:na
        a = arrays{i, 2};
        if ndims(a) == 3
            dims(i, 1:3) = [size(a, 3) size(a, 1) size(a, 2)];
            a = permute(a, [2 1 3]);
        elseif isvector(a)
            dims(i, 1) = numel(a);
        else
            dims(i, 1:2) = size(a);
            a = a.';
        end
        a = a(:);
        if ~arrays{i, 3}; a = max(min(a, 1e20), -1e20); end
        payload{i} = a;
    end
    off = align(128 + 48*na); offs = zeros(na, 1);
    for i = 1:na
        offs(i) = off;
        off = align(off + (8 - 4*arrays{i, 3})*numel(payload{i}));
    end
    total = off;
    f = fopen(path, 'w', 'ieee-le');
    fwrite(f, 'SPCSBLB1', 'char');
    fwrite(f, [1 128 hdr.formulation hdr.method hdr.submethod hdr.flags hdr.n hdr.m hdr.N hdr.k_max na 0], 'uint32');
    fwrite(f, total, 'uint64');
    fwrite(f, [hdr.tol hdr.rho hdr.rho_i hdr.reserved(:).'], 'double');
    for i = 1:na
        fwrite(f, [arrays{i, 1} arrays{i, 3}], 'uint32');
        fwrite(f, [offs(i) numel(payload{i})], 'uint64');
        fwrite(f, [dims(i, :) 0 0], 'uint32');
    end
    for i = 1:na
        fwrite(f, zeros(offs(i) - ftell(f), 1), 'uint8');
        if arrays{i, 3}; fwrite(f, payload{i}, 'int32'); else; fwrite(f, payload{i}, 'double'); end
    end
    fwrite(f, zeros(total - ftell(f), 1), 'uint8');
    fclose(f);
end
