"""curvlinops_amd: MI355X-native backend for curvlinops' curvature-matvec hot path."""
