import sys; import os; R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import numpy as np, shape_cases as SC
cases=[('emagls2', 590, 128, 342, 48000.0, 0.05603182265749108, 6, 4, 'real'),
 ('emagls', 486, 64, 248, 96000.0, 0.027496569985500114, 30, 4, 'real'),
 ('emagls2', 165, 64, 224, 48000.0, 0.03608547132175317, 30, 4, 'complex'),
 ('emainch', 438, 128, 336, 32000.0, 0.06818923842651228, 13, 2, 'complex'),
 ('emagls', 500, 64, 128, 48000.0, 0.08, 32, 4, 'complex'),
 ('emagls', 300, 64, 128, 48000.0, 0.042, 32, 4, 'real')]
from emagls_amd._lib import EmaglsError
for c in cases:
    try: print(c, 'rel=%.2e' % SC.run(c))
    except EmaglsError as e: print(c, 'refused', str(e)[:120])
