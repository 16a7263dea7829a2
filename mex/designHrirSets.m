function [wL, wR] = designHrirSets(kind, hL, hR, hrirGridAziRad, hrirGridZenRad, micRadius, micGridAziRad, micGridZenRad, order, fs, len, shDefinition)
% [wL, wR] = designHrirSets(kind, hL, hR, hrirGridAziRad, hrirGridZenRad, micRadius, micGridAziRad, micGridZenRad, order, fs, len, shDefinition)
%
% The loop over HRIR sets (subjects) around one of the toolbox's design functions as ONE call into the MI355X library:
%
%     for i = 1:size(hL, 3)
%         [wL(:,:,i), wR(:,:,i)] = getEMagLsFilters(hL(:,:,i), hR(:,:,i), hrirGridAziRad, hrirGridZenRad, ...
%                                                   micRadius, micGridAziRad, micGridZenRad, order, fs, len, shDefinition);
%     end
%
% kind           .. 'ls' | 'magls' | 'magls2d' | 'emagls' | 'emagls2' | 'emainch' | 'emainsh'  (getLsFilters, getMagLsFilters,
%                   getMagLsFilters2D, getEMagLsFilters, getEMagLs2Filters, getEMagLsFiltersEMAinCH, getEMagLsFiltersEMAinSH)
% hL, hR         .. [numSamples x numDirections x numSets], all sets on the same HRIR grid (and the same array)
% hrirGridZenRad .. [] for 'magls2d'; micRadius, micGridAziRad, micGridZenRad .. [] where the single call has no such argument
%                   (micGridZenRad also for 'emainch' / 'emainsh'); fs, len .. ignored for 'ls'
% wL, wR         .. [len x numChannels x numSets]: the filters the single calls return
%
% The geometry-only stages (SH matrices, array model, every bin's regularised inverse) run once per batch of up to 16 sets and
% the sequential MagLS sweep is one resident launch per batch (emagls_design_hrir_sets in include/emagls.h).
if nargin < 12 || isempty(shDefinition), shDefinition = 'real'; end
[wL, wR] = emagls_mex('sets', kind, double(hL), double(hR), double(hrirGridAziRad(:)), double(hrirGridZenRad(:)), micRadius, ...
                      double(micGridAziRad(:)), double(micGridZenRad(:)), order, fs, len, shDefinition);
end
