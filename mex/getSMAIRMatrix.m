function [smairMat, params] = getSMAIRMatrix(params)
% dependencies/getSMAIRMatrix.m:1 on the MI355X library: the array model materialised (plane-wave model, rigid sphere, built-in
% getSH; the defaults of :37-84).  The filter-design wrappers never call this: the library works on the model's factors.
% the reference's own defaults (:36-84), 'regul' included: getRadialFilter.m:63-64 rejects it, so -- like there -- a call for the
% SH-domain model has to name its radial filter
if ~isfield(params, 'order'); params.order = 4; end
if ~isfield(params, 'fs'); params.fs = 48000; end
if ~isfield(params, 'smaRadius'); params.smaRadius = 0.042; end
if ~isfield(params, 'oversamplingFactor'); params.oversamplingFactor = 4; end
if ~isfield(params, 'irLen'); params.irLen = 2048; end
if ~isfield(params, 'shDefinition'); params.shDefinition = 'real'; end
if ~isfield(params, 'returnRawMicSigs'); params.returnRawMicSigs = false; end
if ~isfield(params, 'radialFilter'); params.radialFilter = 'regul'; end
if ~isfield(params, 'regulConst'); params.regulConst = 1e-2; end
if ~isfield(params, 'noiseGainDb'); params.noiseGainDb = 20; end
if params.returnRawMicSigs; params.radialFilter = 'none'; end   % (:124-126: never looked at for raw microphone signals)
if ~any(strcmpi(params.radialFilter, {'none', 'tikhonov', 'softlimit', 'full'})); error('Unkown radialFilter parameter "%s".', params.radialFilter); end
if isfield(params, 'shFunction') && ~isequal(func2str(params.shFunction), 'getSH'); error('eMagLS:arg', 'only the built-in getSH is accelerated'); end
if isfield(params, 'arrayType') && ~strcmpi(params.arrayType, 'rigid'); error('eMagLS:arg', 'only the rigid-sphere model is accelerated'); end
smairMat = emagls_mex('smair', params.order, params.fs, params.irLen, params.oversamplingFactor, params.smaRadius, ...
    double(params.smaDesignAziZenRad(:,1)), double(params.smaDesignAziZenRad(:,2)), params.shDefinition, params.returnRawMicSigs, ...
    params.radialFilter, params.regulConst, params.noiseGainDb);
end
