#!/bin/sh
# resetCustomDB.sh — forgets everything derived from <dir>/Custom after its sequences changed (CuCLARK's
# resetCustomDB.sh:24-52): targets.txt, the custom k-mer databases and the custom metadata, then lists the sequences again.
if [ "$1" = "--help" ]; then
  echo "This script erases all database files created with old Custom sequences."
  echo "Please use this script after having updated the Custom folder."
  exit 0
fi
if [ ! -s ./.DBDirectory ]; then
  echo "There is no database directory: run set_targets.sh first."
  exit 1
fi
echo "Are you sure you have updated the Custom directory ? (yes/no)"
read -r decision
case "$decision" in
  yes|y|Y|Yes|YES) ;;
  *) exit 0 ;;
esac
while read -r DIR; do
  [ -n "$DIR" ] && [ -d "$DIR" ] || continue
  printf "The program will clean all database files created with the previous data in the Custom directory..."
  [ -f "$DIR/targets.txt" ] && unlink "$DIR/targets.txt"
  # the custom k-mer databases (custom_<rank>_canonical, <db>_custom_<rank>_canonical) and the custom metadata
  find "$DIR" -mindepth 1 -maxdepth 1 \( -name 'custom*' -o -name '*_custom*' -o -name '.custom*' \) ! -name Custom \
       -exec sh -c 'if [ -d "$1" ]; then find "$1" -delete; else unlink "$1"; fi' _ {} \;
  echo "done"
  printf "Resetting the list of custom sequences..."
  find "$DIR/Custom/" -name '*.f*' > "$DIR/.custom"
  echo "done"
done < ./.DBDirectory
