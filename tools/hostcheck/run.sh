#!/bin/bash
# Host-side sanitizer runs of the Rosenbrock attempt templates (see ros_host_check.cpp).  No GPU, no HIP runtime.
#   tools/hostcheck/run.sh > profiles/r6/ros_host_check.txt
set -u
cd "$(dirname "$0")"
CLANG=/opt/rocm/lib/llvm/bin/clang++
W="-Wno-attributes -Wno-unknown-pragmas -Wno-ignored-attributes -Wno-unknown-attributes"
echo "# (1) g++ -O1 -fsanitize=address,undefined -fno-sanitize-recover=undefined -ffp-contract=off"
g++ -std=c++17 -O1 -g -ffp-contract=off -fsanitize=address,undefined -fno-sanitize-recover=undefined $W ros_host_check.cpp -o /tmp/ros_host_asan && /tmp/ros_host_asan; echo "exit $?"
echo "# (2) clang++ -O3 -fsanitize=address,undefined (the device compiler's front end and optimiser, host target)"
$CLANG -std=c++17 -O3 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined $W ros_host_check.cpp -o /tmp/ros_host_asan_clang && /tmp/ros_host_asan_clang; echo "exit $?"
for init in pattern zero; do
  echo "# (3) clang++ -O3 -ftrivial-auto-var-init=$init"
  $CLANG -std=c++17 -O3 -ftrivial-auto-var-init=$init $W ros_host_check.cpp -o /tmp/ros_host_$init && /tmp/ros_host_$init | tail -1; echo "exit $?"
done
echo "# (3b) the PRODUCT's forms (rolled attempt and run-time indexed pivot exchanges from 13 states on): same digest"
g++ -std=c++17 -O1 -g -ffp-contract=off -DPCG_ROS_ROLLED_ABOVE=12 -fsanitize=address,undefined -fno-sanitize-recover=undefined $W ros_host_check.cpp -o /tmp/ros_host_prod && /tmp/ros_host_prod | tail -1; echo "exit $?"
echo "# (4) g++ -O2 -Wall -Wextra -Wuninitialized -Wmaybe-uninitialized -Warray-bounds=2: diagnostics in the product headers"
g++ -std=c++17 -O2 -Wall -Wextra -Wuninitialized -Wmaybe-uninitialized -Warray-bounds=2 $W -Wno-unused-parameter -Wno-unused-variable -Wno-unused-function ros_host_check.cpp -o /tmp/ros_host_warn 2>&1 | grep -E "warning" | grep -E "uninit|array-bounds|overflow" | sort | uniq -c | head -20
echo "(end of diagnostics)"
