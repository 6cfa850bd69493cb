#!/bin/bash
# round 5, call 26 (three runs, the switches were removed again): HIP stream priorities (this stack offers 0 and -1 = high).
# (a) the weight-gradient stream at -1: 33.73-33.74 ms per step against 33.45-33.52;
# (b) the launch stream of the step at -1 (torch.cuda.set_stream of a priority stream before anything is allocated): 34.02-34.09 against
#     34.13-34.19 on one box, 34.36-34.54 against 34.44-34.61 on another -- at most 0.1 ms, not kept.
