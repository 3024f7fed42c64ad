#!/usr/bin/env python3
"""
Spot finding over directory trees of TIFF images, on MI355X GPUs.

Every *.tif file below the given directories goes through pflib.parallel_image_batch; per image a pickle and a tab-separated
table of its PSFs are written next to it (pflib.save_psfs_pkl / pflib.save_psfs_csv).

Drop-in for the reference's basic_image_script.py (basic_image_script.py:36-124: same options, same directory walk, same call
into pflib, same log lines).  One GPU:
    python -m fluorosequencingimageanalysis_amd.basic_image_script [options] DIR [DIR ...]
all GPUs of a node (the ranks take the place of the reference's worker processes):
    python -m torch.distributed.run --standalone --nproc-per-node 8 \\
        -m fluorosequencingimageanalysis_amd.basic_image_script [options] DIR [DIR ...]
"""
import argparse
import ast
import datetime
import logging
import os
import sys
import time

from . import distributed, pflib


class _Formatter(argparse.ArgumentDefaultsHelpFormatter, argparse.RawDescriptionHelpFormatter):
    pass


def build_parser(timestamp_datetime):
    """The reference's command line (basic_image_script.py:36-83)."""
    p = argparse.ArgumentParser(description=__doc__, formatter_class=_Formatter)
    p.add_argument('--parameters', type=str, nargs=1, default=[None],
                   help="Keyword arguments for pflib.find_peptides as a quoted Python dict literal, e.g. "
                        "--parameters=\"{'median_filter_size': 6, 'c_std': 3}\"; whatever is not named keeps its default.")
    p.add_argument('-mc', '--monte_carlo', action='store_true', default=False,
                   help="Use Monte Carlo method to peakfit (not reproduced on the GPU: find_peptides raises "
                        "NotImplementedError for every image, which is logged like any per-image failure).")
    p.add_argument('--N_iter', type=int, nargs=1, default=[10**3], help="Sample count handed to find_peptides together with --monte_carlo.")
    p.add_argument('-n', '--num_processes', type=int, nargs=1, default=[None],
                   help="Number of processes to use (validated like the reference; the parallel workers here are the "
                        "ranks of the torch.distributed job).")
    default_log = os.path.join('/home', 'basic_image_script_' + str(timestamp_datetime) + '.log')
    p.add_argument('-L', '--log_path', nargs=1, default=[default_log],
                   help="Log file (appended to when it exists).")
    p.add_argument('target_directories', nargs='+', help="One or more directories whose trees are searched for *.tif files.")
    return p


def find_target_images(target_directories):
    """All *.tif files under the directories, in os.walk order (basic_image_script.py:107-113)."""
    out = []
    for target_dir in target_directories:
        for root, _subfolders, files in os.walk(target_dir):
            for f in files:
                if f[-4:] == '.tif':
                    out.append(os.path.join(root, f))
    return out


def main(argv=None):
    timestamp_epoch = time.time()
    timestamp_datetime = datetime.datetime.fromtimestamp(timestamp_epoch)
    args = build_parser(timestamp_datetime).parse_args(argv)
    target_directories = [os.path.abspath(d) for d in args.target_directories]
    if int(os.environ.get("WORLD_SIZE", "1")) == 1:
        # the host worker processes first: they import their modules while this process initialises torch / the GPU
        # - and so does the GPU side, on a thread: importing torch, the HIP context, loading the library
        if pflib.prestart_io_workers(len(find_target_images(target_directories)), args.num_processes[0]) is not None:
            pflib.start_gpu_warmup()
    rank, world, _local = distributed.init_from_env()
    log_path = args.log_path[0] if rank == 0 else args.log_path[0] + '.rank%d' % rank
    logging.basicConfig(filename=log_path, level=logging.DEBUG, force=True)
    logger = logging.getLogger()
    logger.info("basic_image_script starting at " + str(timestamp_datetime))
    fp_parameters = ast.literal_eval(args.parameters[0]) if args.parameters[0] is not None else None
    if args.monte_carlo:
        if fp_parameters is None:
            fp_parameters = {}
        fp_parameters.setdefault('fit_type', 'monte_carlo')
        fp_parameters.setdefault('N_iter', args.N_iter[0])
    target_images = find_target_images(target_directories)
    if world > 1:               # one file list and one timestamp for every rank
        import torch.distributed as dist
        box = [target_images, timestamp_epoch]
        dist.broadcast_object_list(box, src=0)
        target_images, timestamp_epoch = box
    logger.info("Scanned target directories\n" + '\n'.join(target_directories))
    logger.info("Will process target images\n" + '\n'.join(target_images))
    processed_images = pflib.parallel_image_batch(target_images, find_peptides_parameters=fp_parameters,
                                                  timestamp_epoch=timestamp_epoch, num_processes=args.num_processes[0])
    logger.info("Pathnames of images processed: " + str('\n'.join(processed_images.keys())))
    logger.info("basic_image_scipt finished at " + str(datetime.datetime.now()))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return processed_images


if __name__ == "__main__":
    try:
        main()
    except BaseException as e:      # noqa: BLE001
        if isinstance(e, SystemExit):
            raise
        # under torchrun a failed rank must not leave its peers waiting in a collective: report and leave without the
        # interpreter's orderly shutdown (which would block in the process group's destructor); torchrun then ends the job
        import traceback
        traceback.print_exc()
        logging.getLogger().exception(e)
        logging.shutdown()
        sys.stderr.flush()
        os._exit(1) if distributed.world_size() > 1 else sys.exit(1)
    sys.exit(0)
