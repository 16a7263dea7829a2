# accuracy probe of fast_rcp / fast_rsqrt through the Jacobi: singular values vs LAPACK on a thin design
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
