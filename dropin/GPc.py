"""``import GPc`` resolves to the MI355X-native classification module (the reference's GPc.py does not parse)."""
from gptest_amd.GPc import *  # noqa: F401,F403
from gptest_amd.GPc import ClassifierLikelihood, ClassificationGaussianProcess, NormCDF, logistic_function  # noqa: F401
