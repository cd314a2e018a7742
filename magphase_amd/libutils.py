"""
Python-3 counterpart of the helper names the reference's scripts import from src/libutils.py
(SURVEY.md section 8b).  Only I/O and glue: no signal arithmetic lives here.
"""
import os
import socket
from multiprocessing import Pool

import numpy as np

from .hostmath import round_to_int  # noqa: F401  (libutils.py:131-133)


def read_binfile(filename, dim=60):
    """libutils.py:112-120: raw little-endian float32, row-major, no header -> float64, squeezed."""
    v_data = np.fromfile(filename, dtype=np.float32)
    if np.mod(v_data.size, dim) != 0:
        raise ValueError("Dimension provided not compatible with file size.")
    return np.squeeze(v_data.reshape((-1, dim)).astype("float64"))


def write_binfile(m_data, filename):
    """libutils.py:122-127."""
    np.array(m_data, "float32").tofile(filename)


def read_text_file2(*args, **kargs):
    """libutils.py:99-102.  dtype='string' (numpy 1 / python 2 spelling used by the scripts) maps to str."""
    if kargs.get("dtype", None) == "string":
        kargs["dtype"] = str
    return np.atleast_1d(np.genfromtxt(*args, **kargs))


def read_scp_file(filename):
    """libutils.py:94-95."""
    return read_text_file2(filename, dtype="string", comments="#")


def get_filename(filepath):
    """libutils.py:142-144."""
    return os.path.splitext(os.path.basename(filepath))[0]


def mkdir(l_dir):
    """libutils.py:146-156.  Several ranks of one batch job call this on the same path at the same moment: creation is
    race-free (a directory another rank made in between is not an error)."""
    if isinstance(l_dir, str):
        l_dir = [l_dir]
    for directory in l_dir:
        try:
            os.mkdir(directory)
        except FileExistsError:
            if not os.path.isdir(directory):
                raise


def ins_pid(filepath):
    """libutils.py:187-195: path/file.ext -> path/file_<host>_<pid>.ext."""
    filename, ext = os.path.splitext(filepath)
    return "%s_%s_%d%s" % (filename, socket.gethostname(), os.getpid(), ext)


def _func_wrapper(args):
    args[0](*args[1:])


def run_multithreaded(*args):
    """
    libutils.py:32-63: the reference's only parallelism -- one utterance per Pool worker.  Kept for script
    compatibility; on the GPU path utterances are batched per device instead (magphase_amd/sharding.py).
    """
    func = args[0]
    nruns = next(len(a) for a in args[1:] if type(a) is list)
    jobs = [tuple([func] + [a[i] if type(a) is list else a for a in args[1:]]) for i in range(nruns)]
    with Pool() as pool:
        return pool.map(_func_wrapper, jobs)


# ---- small helpers of the reference's libutils kept for scripts that import them (libutils.py:23-30, 66-95, 129-203)
def fileparts(fullpath):
    """libutils.py:135-139 -> [directory, file token, extension, path without extension]."""
    path_no_ext, ext = os.path.splitext(fullpath)
    return [os.path.dirname(fullpath), os.path.basename(path_no_ext), ext, path_no_ext]


def get_file_list(files_path):
    """libutils.py:106-109: glob pattern -> (list, count)."""
    import glob
    files = glob.glob(files_path)
    return files, len(files)


def gen_list_of_file_paths(files_dir, v_file_tkns, suffix):
    """libutils.py:66-77."""
    return [files_dir + "/" + str(t) + suffix for t in v_file_tkns]


def indexes_to_one_zero_vector(v_nxs, length):
    """libutils.py:82-91."""
    v = np.zeros(length)
    v[np.asarray(v_nxs).astype(int)] = 1
    return v


def is_mutable(data):
    return hasattr(data, "__setitem__")


def add_rel_path(rel_path):
    """libutils.py:175-180: appends <directory of the calling file> + rel_path to sys.path."""
    import inspect
    import sys
    caller_dir = os.path.dirname(inspect.stack()[1][1])
    sys.path.append(os.path.realpath(caller_dir + rel_path))


def ins_date_time(filepath, prefix=""):
    """libutils.py:197-203: path/file.ext -> path/file_<prefix>_<YYYYmmdd_HHMM>.ext."""
    import time
    name, ext = os.path.splitext(filepath)
    return "%s_%s_%s%s" % (name, prefix, time.strftime("%Y%m%d_%H%M"), ext)


func_wrapper = _func_wrapper
