function sigFiltered = applyRadialFilter(inSig, params)
% dependencies/applyRadialFilter.m:1 on the MI355X library; params.nfft must be oversamplingFactor * irLen (verifyEMagLs.m:250)
if ~isfield(params, 'radialFilter'); params.radialFilter = 'tikhonov'; end
if ~isfield(params, 'oversamplingFactor'); params.oversamplingFactor = 2; end
if ~isfield(params, 'irLen'); params.irLen = 256; end
if ~isfield(params, 'regulConst'); params.regulConst = 1e-2; end
if ~isfield(params, 'noiseGainDb'); params.noiseGainDb = NaN; end
assert(params.nfft == params.oversamplingFactor * params.irLen, 'params.nfft must equal oversamplingFactor * irLen');
if size(inSig, 1) < params.nfft; disp('applyRadialFilter: short signal, applying zero padding!'); end
sigFiltered = emagls_mex('applyradial', double(inSig), params.order, params.fs, params.smaRadius, params.irLen, ...
    params.oversamplingFactor, params.radialFilter, params.regulConst, params.noiseGainDb);
end
