function [wMlsL, wMlsR] = getEMagLsFiltersFromAtf(hL, hR, hrirGridAziZenRad, atfIrs, atfGridAziZenRad, fs, filterLen, fTrans)
[wMlsL, wMlsR] = emagls_mex('fromatf', double(hL), double(hR), double(hrirGridAziZenRad), double(atfIrs), double(atfGridAziZenRad), fs, filterLen, fTrans);
end
