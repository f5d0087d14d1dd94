class Config(object):
    use_openmp = False
    use_opencl = False
    use_cuda = False
    use_double = True
    use_local_memory = False
    omp_schedule = ('dynamic', 64)
    profile = False
    wgs = 32


_cfg = Config()


def get_config():
    return _cfg
