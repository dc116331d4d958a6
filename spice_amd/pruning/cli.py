"""`python -m spice_amd.pruning.cli --model M --strategy S --ratio R [...]`
(pruning/cli.py:16-72: same flags, exit status 1 on any failure, same summary block)."""
from __future__ import annotations

import logging
import sys

from .cli_helpers import build_config_from_args, create_parser, validate_args
from .utils import setup_logging

logger = logging.getLogger(__name__)


def parse_args(argv=None):
    args = create_parser().parse_args(argv)
    validate_args(args)
    return args


def _print_summary(r) -> None:
    bar = "=" * 60
    print("\n" + bar + "\nPRUNING COMPLETE\n" + bar)
    print(f"Model: {r['model_name']}")
    print(f"Strategy: {r['pruning_strategy']}")
    if r.get("kd_lite_enabled"):
        print(f"KD-Lite: Enabled (T={r['kd_temperature']}, α={r['kd_alpha']})")
    print(f"Parameters: {r['initial_params']:,} → {r['final_params']:,}")
    print(f"Reduction: {r['params_reduction']:.2%}")
    print(f"Model size: {r['initial_size_mb']:.2f} MB → {r['final_size_mb']:.2f} MB")
    print(f"Size reduction: {r['size_reduction']:.2%}")
    print(f"Output: {r['output_path']}")
    print(bar)


def main(argv=None) -> None:
    setup_logging()
    try:
        args = parse_args(argv)
        config = build_config_from_args(args)
        # under torch.distributed.run: one process per GPU, rank-local device
        from .. import dist as sdist
        rank, world, local = sdist.init_from_env()
        if world > 1 and config.device == "cuda":
            config.device = f"cuda:{local}"
        logger.info(f"Starting structured pruning: model={config.model_name} strategy={config.pruning_strategy} "
                    f"ratio={config.pruning_ratio} steps={config.iterative_steps} device={config.device} "
                    f"output={config.output_dir}")
        if config.enable_kd_lite:
            logger.info(f"  KD-Lite: T={config.kd_temperature} alpha={config.kd_alpha} data={config.kd_data_ratio}")
        from .pruner import prune_model
        results = prune_model(config)
        if rank == 0:
            _print_summary(results)
    except KeyboardInterrupt:
        logger.info("Pruning interrupted by user")
        sys.exit(1)
    except Exception as e:
        logger.error(f"Pruning failed: {e}")
        sys.exit(1)


if __name__ == "__main__":
    main()
