#!/bin/bash
# s31: the round's profile set on the final build
bash tools/profile_round.sh r04d > /dev/null 2>&1
