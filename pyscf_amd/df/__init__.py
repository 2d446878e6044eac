from . import addons, df_jk
from .addons import make_auxmol
from .df import DF, GDF
