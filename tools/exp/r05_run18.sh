#!/bin/bash
# round 5, call 18 (two runs): WHEN the fc2 / fc1 weight gradients of a block reach the side stream -- (a) in front of the block's attention
# backward instead of right behind the GELU' product: 34.37-34.42 vs 34.07-34.14 ms per step; (b) fc2's in front of the GELU' product (its
# operands exist since the end of the block before), fc1's where it is: 34.02-34.03 vs 33.92-34.06.  Neither kept; the engine switches
# (wgrad_mlp_late / wgrad_fc2_early, bench --wgrad-mlp-late / --wgrad-fc2-early) were removed again.
