function binauralOut = binauralDecode(in, inFs, decodingFilterLeft, decodingFilterRight, decodingFilterFs, compensateDelay, signal, signalFs, horRotAngleRad)
% Core loop of the reference's binauralDecode (dependencies/binauralDecode.m:33-64) on the GPU, real or complex SH signals and
% filters; resampling, rotation and the extra convolution stay in MATLAB.
if decodingFilterFs ~= inFs
    decodingFilterLeft = resample(decodingFilterLeft, inFs, decodingFilterFs);
    decodingFilterRight = resample(decodingFilterRight, inFs, decodingFilterFs);
end
if nargin > 8 && ~isempty(horRotAngleRad) && horRotAngleRad ~= 0; in = rotateHOA_N3D(in, rad2deg(horRotAngleRad), 0, 0); end
comp = nargin > 5 && compensateDelay;
if isreal(decodingFilterLeft) ~= isreal(decodingFilterRight)      % both real or both complex at the boundary
    decodingFilterLeft = complex(decodingFilterLeft); decodingFilterRight = complex(decodingFilterRight);
end
extraConv = nargin > 6 && ~isempty(signal);
% the library cuts the delay itself (and sums the discarded imaginary part over the samples it returns, like :53-62) unless the
% extra convolution of :44-48 has to run on the uncut signal first
[binauralOut, imagSum] = emagls_mex('decode', double(in), double(decodingFilterLeft), double(decodingFilterRight), comp && ~extraConv);
if extraConv
    if signalFs ~= inFs; signal = resample(signal, inFs, signalFs); end
    binauralOut = [fftfilt(binauralOut(:,1), signal(:,1)), fftfilt(binauralOut(:,2), signal(:,1))];
    if comp; del = size(decodingFilterLeft,1) / 2; binauralOut = binauralOut(del:end,:); end
end
% :59-63, the reference's text; it fires when the accumulated result is complex, i.e. has a non-zero imaginary part
if (~isreal(in) || ~isreal(decodingFilterLeft)) && any(imagSum ~= 0)
    warning('discarding imaginary part with sum of [%.2g, %.2g] in rendering result.', imagSum(1), imagSum(2));
end
end
