#!/usr/bin/env python3
"""Regenerate sba-gan_amd/cfg/*.yml from the VALUES of the reference's nine yml files
(/root/reference/AttnGAN2/code/cfg), so that yaml.safe_load of every shipped file equals the reference's.
Only parsed values are carried over (keys in the reference's order, no comments or layout);
tests/test_host_cpu.py::test_shipped_cfg_equals_reference holds the result to it.  Build container only."""
import os

import yaml

REF = '/root/reference/AttnGAN2/code/cfg'
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'sba-gan_amd', 'cfg')


def main():
    n = 0
    for dp, _, fs in os.walk(REF):
        for f in sorted(fs):
            if not f.endswith('.yml'):
                continue
            rel = os.path.relpath(os.path.join(dp, f), REF)
            with open(os.path.join(dp, f)) as fh:
                d = yaml.safe_load(fh)
            dst = os.path.join(OUT, rel)
            os.makedirs(os.path.dirname(dst), exist_ok=True)
            with open(dst, 'w') as fh:
                fh.write('# %s: values of the reference configuration of the same name (regenerate: tools/make_cfg.py)\n' % rel)
                yaml.safe_dump(d, fh, default_flow_style=False, sort_keys=False, indent=4)
            n += 1
    print('wrote %d files' % n)


if __name__ == '__main__':
    main()
