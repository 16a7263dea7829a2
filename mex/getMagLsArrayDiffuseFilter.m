function wAdf = getMagLsArrayDiffuseFilter(micRadius, micGridAziRad, micGridZenRad, order, fs, len, shDefinition, shFunction)
% lib/getMagLsArrayDiffuseFilter.m:1 on the MI355X library; a custom shFunction is evaluated here at ceil(fs*pi*micRadius/343) (:38)
if nargin < 7 || isempty(shDefinition); shDefinition = 'real'; end
if nargin >= 8 && ~isequal(func2str(shFunction), 'getSH')
    Yhi = shFunction(ceil(fs * pi * micRadius / 343), [micGridAziRad(:), micGridZenRad(:)], shDefinition);
    wAdf = emagls_mex('adf', micRadius, double(micGridAziRad(:)), double(micGridZenRad(:)), order, fs, len, shDefinition, Yhi);
    return;
end
wAdf = emagls_mex('adf', micRadius, double(micGridAziRad(:)), double(micGridZenRad(:)), order, fs, len, shDefinition);
end
