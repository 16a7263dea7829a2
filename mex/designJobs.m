function W = designJobs(jobs, batchSize, inFlight, shareGeometry, devices)
% W = designJobs(jobs, batchSize, inFlight, shareGeometry, devices)
%
% A list of independent filter designs in ONE call on the MI355X library: the loop a script writes around
% getEMagLsFilters / getEMagLs2Filters / getMagLsFilters / getEMagLsFiltersFromAtf ... (testEMagLs.m:75-95 over array radii,
% testEMagLsFromAtfs.m:72-73 over subjects).  The library cuts the list into chunks of one shape, runs every chunk as one
% batch (one launch of each kernel for all its designs, one resident sweep) and keeps several chunks in flight.
%
% jobs          .. struct array, one element per design, fields named like the reference's arguments:
%                  kind ('ls','magls','magls2d','emagls','emagls2','emainch','emainsh','fromatf'), hL, hR,
%                  hrirGridAziRad, hrirGridZenRad, order, fs, len, shDefinition, micRadius, micGridAziRad, micGridZenRad,
%                  atfIrs, atfGridAziRad, atfGridZenRad, fTrans, applyDiffusenessConst, simOrderPad
% batchSize     .. designs per chunk (default 32), inFlight .. chunks in flight (default 4)
% shareGeometry .. true: designs of a chunk that differ only in their HRIRs compute the geometry stages once
% devices       .. GPUs of this MATLAB process to split the list over, e.g. 0:7 (default: the current device).  Array-radius
%                  studies are cut into lane batches of equal cost and whole batches go to devices (emagls_jobs_shard); every
%                  device runs its share from a host thread of its own and writes its filters straight into W: no gather
% W             .. numel(jobs) x 2 cell array {wL, wR}: the filters each single call returns
%
% Example (256 array radii, BASELINE config 4):
%   for i = 1:numel(radii)
%       jobs(i) = struct('kind','emagls2','hL',hL,'hR',hR,'hrirGridAziRad',azi,'hrirGridZenRad',zen,'micRadius',radii(i), ...
%                        'micGridAziRad',micAzi,'micGridZenRad',micZen,'order',4,'fs',48000,'len',1024,'shDefinition','real');
%   end
%   W = designJobs(jobs);
if nargin < 2, batchSize = []; end
if nargin < 3, inFlight = []; end
if nargin < 4, shareGeometry = false; end
if nargin < 5, devices = []; end
W = emagls_mex('jobs', jobs, batchSize, inFlight, logical(shareGeometry), double(devices));
end
