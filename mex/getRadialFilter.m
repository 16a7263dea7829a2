function radFilts = getRadialFilter(params)
% dependencies/getRadialFilter.m:1 on the MI355X library (plane-wave model, rigid sphere); the defaults of :27-41,58-60
if ~isfield(params, 'radialFilter'); params.radialFilter = 'tikhonov'; end
if ~isfield(params, 'waveModel'); params.waveModel = 'planeWave'; end
if ~isfield(params, 'oversamplingFactor'); params.oversamplingFactor = 2; end
if ~isfield(params, 'irLen'); params.irLen = 256; end
if ~isfield(params, 'dirCoeff'); params.dirCoeff = 0; end
if ~isfield(params, 'regulConst'); params.regulConst = 1e-2; end
if ~isfield(params, 'noiseGainDb'); params.noiseGainDb = NaN; end
if ~strcmpi(params.radialFilter, 'none')
    if strcmpi(params.waveModel, 'pointSource'); error('WaveModel parameter "%s" not yet implemented.', params.waveModel); end
    if ~strcmpi(params.arrayType, 'rigid') || params.dirCoeff ~= 0; error('eMagLS:arg', 'only the rigid-sphere model is accelerated'); end
end
radFilts = emagls_mex('radial', params.order, params.fs, params.smaRadius, params.irLen, params.oversamplingFactor, ...
    params.radialFilter, params.regulConst, params.noiseGainDb);
end
